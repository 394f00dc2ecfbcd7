#!/bin/bash
# One rocprofv3 counter pass over a short bench run on the GPU box; prints per-kernel means.
# Usage: tools/pmc_pass.sh <tag> <counter> [<counter> ...]
set -e -o pipefail
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf "$OUT" && mkdir -p "$OUT/pmc_sq"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_sq" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-match --no-extra --no-stage-times > "$OUT/run.log" 2>&1
python3 "$R/tools/summarise_profiles.py" "$OUT" "$TAG"
