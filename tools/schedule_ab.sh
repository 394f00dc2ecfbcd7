set -e
B="python bench.py --no-match --no-extra --no-cpu --no-stage-times --steps 30 --warmup 5"
for v in "" "ORBX_PARTS=3" "ORBX_PARTS=4" "ORBX_PARTS=1" "ORBX_PYR_WAIT=0" "ORBX_PARTS=3 ORBX_PYR_WAIT=0"; do
  for b in 1024 1536 2048; do
    r=$(env $v timeout -k 10 120 $B --batch $b 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['value']), round(d['ms_per_step'],3))")
    echo "[$v] batch $b: $r"
  done
done
