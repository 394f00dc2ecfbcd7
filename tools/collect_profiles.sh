#!/bin/bash
# Runs on the GPU box: rocprofv3 kernel trace of the default bench command and
# separate counter passes (HBM read bytes, HBM write bytes, SQ issue counters),
# then writes per-kernel summaries.  Usage: tools/collect_profiles.sh <tag>
set -e -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- \
  python3 "$R/bench.py" --steps 20 --warmup 3 --no-cpu --no-extra --no-stage-times > "$OUT/trace_bench.log" 2>&1
echo "trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-match --no-extra --no-stage-times > "$OUT/pmc_fetch.log" 2>&1
echo "fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-match --no-extra --no-stage-times > "$OUT/pmc_write.log" 2>&1
echo "write done"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d "$OUT/pmc_sq" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-match --no-extra --no-stage-times > "$OUT/pmc_sq.log" 2>&1
echo "sq done"
python3 "$R/tools/summarise_profiles.py" "$OUT" "$TAG"
# copy into profiles/ by hand as <tag>_bench_b1024_{kernel_stats.csv,pmc_summary.json}: bench.py quotes the newest
# profiles/r*_bench_b<batch>_pmc_summary.json whose recorded code state matches the running configuration
