#!/bin/bash
# Times the schedule knobs of the batched extraction (parts per batch, where the blur runs, whether
# a part's pyramid waits for the previous part's), one short bench run each.  Usage (repository
# root on the GPU box): tools/ab_schedule.sh <tag>
set -e -o pipefail
TAG=${1:-absched}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
for parts in 2 3; do for blur in 0 1 2; do for wait in 1 0; do
  name=p${parts}_b${blur}_w${wait}
  ORBX_PARTS=$parts ORBX_SIDE_BLUR=$blur ORBX_PYR_WAIT=$wait timeout -k 10 120 python bench.py --no-cpu --no-match --no-extra --steps 20 --warmup 3 > "$OUT/$name.log" 2>&1
  python - "$OUT/$name.log" "$name" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-12s %9.0f frames/s  %.3f ms/step  fast launch %.3f ms" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["launch_ms"]))
PY
done; done; done
