set -e -o pipefail
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" timeout -k 10 120 python bench.py --no-cpu --no-match --no-extra --steps 20 --warmup 3 $EXTRA > gpurun_out/ab2_$name.log 2>&1; python - gpurun_out/ab2_$name.log $name <<'PY'
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("%-16s %9.0f frames/s  %.3f ms/step  fast launch %.3f ms" % (sys.argv[2], j["value"], j["ms_per_step"], j["roofline"]["launch_ms"]))
PY
}
EXTRA="--batch 1536" run b1536_p3 ORBX_PARTS=3
EXTRA="--batch 1536" run b1536_p2 ORBX_PARTS=2
EXTRA="--batch 2048" run b2048_p2 ORBX_PARTS=2
EXTRA="--batch 2048" run b2048_p3 ORBX_PARTS=3
EXTRA="--batch 768" run b768_p3 ORBX_PARTS=3
EXTRA="--batch 512" run b512_p2 ORBX_PARTS=2
