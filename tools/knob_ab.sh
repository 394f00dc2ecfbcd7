#!/bin/bash
# A/B of the timing knobs on the headline loop (device-resident, 1024 frames per step).
# Usage (GPU box): bash tools/knob_ab.sh
B="python bench.py --no-match --no-extra --no-cpu --no-stage-times --steps 30 --warmup 5"
for v in "" "ORBX_PYR_WAVES=2" "ORBX_PYR_WAVES=4" "ORBX_PYR_XCD=0" "ORBX_FAST_QCAP=600" "ORBX_FAST_QCAP=1000" "ORBX_OCT_KEYS_LDS=1100" "ORBX_FAST_SINGLE_PASS=1" ""; do
  r=$(env $v timeout -k 10 120 $B 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['ms_per_step'],3))")
  echo "[$v]: $r"
done
