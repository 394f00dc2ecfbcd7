#!/bin/bash
# Per-kernel durations with every stage alone on the chip: kernel trace of bench.py --stage-times
# (the serialised section launches each stage for the whole batch on one stream).  Prints the mean
# duration of the launches that cover the whole batch.  Usage (GPU box): tools/alone_times.sh <tag>
set -e -o pipefail
TAG=${1:-alone}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
rm -rf "$OUT" && mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- \
  python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu --no-extra --no-match --stage-times > "$OUT/bench.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/trace/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    n = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("orbx::", "").replace("void ", "")
    if not n.startswith("k_"): continue
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    by[n].append((grid, int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
print("%-22s %8s %10s %s" % ("kernel", "launches", "grid", "mean us of the whole-batch launches (per launch position in the chain)"))
for n, v in sorted(by.items()):
    gmax = max(g for g, _, _ in v)
    # whole-batch launches = those with the largest grids; group by grid size
    groups = collections.defaultdict(list)
    for g, s, d in v:
        groups[g].append(d)
    big = sorted(groups.items(), reverse=True)
    txt = ", ".join("%d:%0.1f(x%d)" % (g, sum(d) / len(d) / 1e3, len(d)) for g, d in big[:6])
    print("%-22s %8d %10d %s" % (n[:22], len(v), gmax, txt))
PY
