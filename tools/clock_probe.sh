#!/bin/bash
# The clock the chip holds inside each kernel: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs)
# against the kernel's duration from the same run's kernel trace.  Counter collection runs the
# kernels one at a time, so these are "alone on the chip" clocks.  Usage (GPU box): tools/clock_probe.sh <tag>
TAG=${1:-clock}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 150 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu --no-extra --no-stage-times > "$OUT/run.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
d = sys.argv[1] + "/p"
cnt = {}
for f in glob.glob(d + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Dispatch_Id"]] = float(r["Counter_Value"])
acc = collections.defaultdict(list)
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("orbx::", "").replace("void ", "")
        if not n.startswith("k_") or r["Dispatch_Id"] not in cnt: continue
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        if dur > 20000: acc[n].append(cnt[r["Dispatch_Id"]] / 8.0 / dur)
for n, v in sorted(acc.items()):
    print("%-22s %3d launches  %.2f GHz (busy cycles per XCD / ns)" % (n[:22], len(v), sum(v) / len(v)))
PY
