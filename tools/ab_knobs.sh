#!/bin/bash
# Times the alternatives the library can switch between without a rebuild, one bench run each
# (every run under its own time limit, chained so that nothing follows a failure).  Usage (from
# the repository root on the GPU box): tools/ab_knobs.sh <tag>
set -e -o pipefail
TAG=${1:-ab}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
run() {   # name, then VAR=value pairs
  local name=$1; shift
  env "$@" timeout -k 10 240 python bench.py --no-cpu --no-match --no-extra --stage-times --steps 20 --warmup 3 > "$OUT/$name.log" 2>&1
  python - "$OUT/$name.log" "$name" <<'PY'
import json, sys
line = [l for l in open(sys.argv[1]) if l.startswith("{")][-1]
j = json.loads(line)
st = j.get("stage_ms_per_step_serialised") or (j.get("config") or {}).get("stage_ms_per_step_serialised") or {}
print("%-28s %9.0f %s  %.3f ms/step  %s" % (sys.argv[2], j["value"], j["unit"], j["ms_per_step"], st))
PY
}
run two_pass_default A=1
run single_pass ORBX_FAST_SINGLE_PASS=1
for k in 2048 4096 8192; do run oct_keys_lds_$k ORBX_OCT_KEYS_LDS=$k; done
# the guided matchers with and without a chip-wide launch for the window walks
for g in 0 1; do
  ORBX_GUIDED_GATHER=$g timeout -k 10 240 python bench.py --no-cpu --no-match --no-extra --guided --steps 5 --warmup 2 > "$OUT/guided_gather_$g.log" 2>&1
  python - "$OUT/guided_gather_$g.log" "$g" <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])["guided"]
print("guided gather=%s: last frame %.3f ms, local map %.3f ms, by bow %.3f ms" % (sys.argv[2], j["value"], j["local_map"]["ms"], j["by_bow"]["ms"]))
PY
done
