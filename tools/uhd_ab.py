import sys, os, json
sys.path.insert(0, os.getcwd())
import torch, bench
import orb_slam3_amd as orbx
dev = torch.device("cuda:0")
for b in (64, 96, 128):
    r = bench.uhd_bench(torch, orbx, dev, 0, batch=b, reps=5)
    print(os.environ.get("ORBX_PARTS", "default"), b, round(r["value"]), round(r["ms_per_step"], 3), flush=True)
