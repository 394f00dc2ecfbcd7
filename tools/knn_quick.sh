#!/bin/bash
# The matching tests and the Hamming leg of the bench on the GPU box (one short call while the
# 2-NN kernel is being worked on).  Usage: tools/knn_quick.sh <tag>
TAG=${1:-knn}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
timeout -k 10 400 python -u -m pytest tests/test_gpu_match.py tests/test_zz_gpu_lapping_knn.py tests/test_zzzz_no_alloc.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/${TAG}_tests.log 2>&1
rc=$?; tail -2 gpurun_out/${TAG}_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --no-cpu --no-extra --no-stage-times --steps 2 --warmup 1 > gpurun_out/${TAG}_bench.log 2>&1 || exit 1
tail -1 gpurun_out/${TAG}_bench.log | python3 -c "import json,sys; j=json.loads(sys.stdin.read())['hamming']; print('%.3f T pairs/s  %.1f ms  frac %.3f' % (j['value']/1e12, j['ms'], j['roofline']['frac']))"
