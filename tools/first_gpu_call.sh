#!/bin/bash
# The first GPU call after a stretch of work without one: every step under its own time limit,
# chained so that nothing runs on the GPU after a failure.  Usage (from the repository root on
# the GPU box): tools/first_gpu_call.sh <tag>
set -e -o pipefail
TAG=${1:-first}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
# oldest tests first; the newest kernels (tests/test_zz_gpu_*.py) are collected last
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > "$OUT/tests.log" 2>&1
echo "tests done: $(tail -1 "$OUT/tests.log")"
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1
echo "smoke done"
timeout -k 10 400 python bench.py --stage-times --bow --guided > "$OUT/bench.log" 2>&1
echo "bench done"
tail -c 3000 "$OUT/bench.log"
