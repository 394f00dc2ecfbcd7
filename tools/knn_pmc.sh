#!/bin/bash
# Counter pass over the Hamming leg of the bench (matrix-core 2-NN kernel).  Usage: tools/knn_pmc.sh <tag>
TAG=${1:-knnpmc}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F6F4 SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  --output-format csv -d "$OUT/a" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu --no-extra --no-stage-times > "$OUT/a.log" 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/b" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu --no-extra --no-stage-times > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(sys.argv[1] + "/" + sub + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_knn2" not in r["Kernel_Name"]: continue
            acc[r["Kernel_Name"][:24]][r["Counter_Name"]] += float(r["Counter_Value"])
            n[(r["Kernel_Name"][:24], r["Counter_Name"])] += 1
    for k, v in acc.items():
        print(k, {c: round(x / n[(k, c)] / 1e6, 2) for c, x in v.items()}, "(millions per launch)")
PY
