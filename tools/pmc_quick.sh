#!/bin/bash
# Quick check on the GPU box: bench line plus one HBM-read counter pass.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/quick
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 python3 "$R/bench.py" --steps 10 --warmup 2 --no-cpu --no-match "$@" | tee "$OUT/bench.log"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-match "$@" > "$OUT/pmc_fetch.log" 2>&1
python3 "$R/tools/summarise_profiles.py" "$OUT" quick
