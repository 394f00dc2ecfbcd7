#!/usr/bin/env python3
"""Condenses rocprofv3 output directories (kernel trace + counter passes written by
tools/collect_profiles.sh) into small per-kernel tables: <out>/<tag>_kernel_stats.csv
(copied from the trace) and <out>/<tag>_pmc_summary.json (mean counter value per launch of
each kernel).  FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes, so the
summary lists both the raw value and the doubled one."""
import csv
import glob
import json
import os
import re
import shutil
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("orbx::", "").replace("(anonymous namespace)::", "")
    return name.replace("void ", "").strip()


def main():
    out, tag = sys.argv[1], sys.argv[2]
    for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(out, "%s_kernel_stats.csv" % tag))
    # A kernel may run at several grid sizes in one bench run (half batches on two streams in
    # the timed loop, whole batches in the per-stage timing pass): only the largest grid of each
    # kernel, i.e. the whole-batch launch the bench's per-stage times refer to, is summarised.
    rows = []
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    k = short(row["Kernel_Name"])
                    if k.startswith("k_"):
                        rows.append((k, int(row["Grid_Size"]), row["Counter_Name"], float(row["Counter_Value"])))
    biggest = {}
    for k, g, _, _ in rows:
        biggest[k] = max(biggest.get(k, 0), g)
    summ = {}
    for k, g, c, v in rows:
        if g != biggest[k]:
            continue
        e = summ.setdefault(k, {}).setdefault(c, [0.0, 0])
        e[0] += v
        e[1] += 1
    res = {}
    for k, cs in sorted(summ.items()):
        r = {c: v[0] / v[1] for c, v in cs.items()}
        r["launches_sampled"] = max(v[1] for v in cs.values())
        r["grid_size"] = biggest[k]
        if "FETCH_SIZE" in r:
            r["hbm_read_bytes_raw"] = r["FETCH_SIZE"] * 1024
            r["hbm_read_bytes_x2"] = r["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in r:
            r["hbm_write_bytes"] = r["WRITE_SIZE"] * 1024
        if "SQ_ACTIVE_INST_VALU" in r and r.get("SQ_BUSY_CYCLES"):
            # 4 SIMDs per CU each issue VALU independently; ACTIVE_INST_VALU is in quad-cycles
            r["valu_per_wave"] = r["SQ_INSTS_VALU"] / max(r.get("SQ_WAVES", 1), 1)
        res[k] = r
    # the code state the counters belong to: bench.py only quotes a summary whose state matches
    # the configuration it is running
    res["_meta"] = {"fast_two_pass": 0 if os.environ.get("ORBX_FAST_SINGLE_PASS") == "1" else 1,
                    "command": "python3 bench.py --steps 2 --warmup 1 --no-cpu --no-match --no-extra "
                               "(one rocprofv3 --pmc pass per counter group)"}
    with open(os.path.join(out, "%s_pmc_summary.json" % tag), "w") as fh:
        json.dump(res, fh, indent=1, sort_keys=True)
    for k, r in res.items():
        if k.startswith("_"):
            continue
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in r.items()})


if __name__ == "__main__":
    main()
