"""world_size-2 (and 3) gloo run of the sharded 2-NN exchange on CPU tensors.  The per-shard
search and the merge are injected from the oracle / numpy so that the product's sharding,
offset and all-gather logic runs without a GPU; on a GPU box the same function runs the HIP
kernels (tests/test_gpu_match.py covers those)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def merge_top2_numpy(idx_parts, dist_parts):
    P, nq, _ = idx_parts.shape
    key = (dist_parts.astype(np.uint64) << np.uint64(32)) | idx_parts.astype(np.uint64)
    key = np.transpose(key, (1, 0, 2)).reshape(nq, 2 * P)
    key.sort(axis=1)
    best = key[:, :2]
    return (best & np.uint64(0xFFFFFFFF)).astype(np.uint32), (best >> np.uint64(32)).astype(np.uint16)


def _worker(rank, world, port, q, t, out, fail_rank=-1):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import orb_slam3_amd as orbx
    from tests import oracle_lib as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_search(qt, tt, off):
        if rank == fail_rank:
            raise orbx.OrbxError(-6, "injected failure of this rank's search")
        i, d = O.knn2(qt.numpy(), tt.numpy())
        i = np.where(i != 0xFFFFFFFF, i + np.uint32(off), i)
        return torch.from_numpy(i.view(np.int32).copy()), torch.from_numpy(d.view(np.int16).copy())

    def merge(ia, da):
        i, d = merge_top2_numpy(ia.numpy().view(np.uint32), da.numpy().view(np.uint16))
        return torch.from_numpy(i.view(np.int32).copy()), torch.from_numpy(d.view(np.int16).copy())

    b = orbx.shard_bounds(len(t), world)
    shard = torch.from_numpy(t[b[rank]:b[rank + 1]].copy())
    try:
        i, d = orbx.knn_match2_sharded(torch.from_numpy(q), shard, b[rank], _hooks=(local_search, merge))
        out[rank] = (i.numpy().view(np.uint32).copy(), d.numpy().view(np.uint16).copy())
    except orbx.OrbxError as e:
        out[rank] = ("error", int(e.status))
    # the group is still usable after a failed search: a second, healthy exchange completes
    if fail_rank >= 0:
        fail_rank = -1
        i, d = orbx.knn_match2_sharded(torch.from_numpy(q), shard, b[rank], _hooks=(local_search, merge))
        out[(rank, "again")] = (i.numpy().view(np.uint32).copy(), d.numpy().view(np.uint16).copy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_knn_equals_unsharded(oracle, world):
    import torch.multiprocessing as mp
    rng = np.random.default_rng(world)
    t = rng.integers(0, 256, (2001, 32), dtype=np.uint8)
    t = np.repeat(t[:400], 6, axis=0)[:2001]          # duplicates across shard boundaries
    q = t[rng.integers(0, len(t), 150)].copy()
    q[::3] ^= rng.integers(0, 256, (50, 32), dtype=np.uint8) & 0x11
    ridx, rdist = oracle.knn2(q, t)
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, t, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        assert np.array_equal(out[r][0], ridx) and np.array_equal(out[r][1], rdist)


def test_sharded_knn_failure_on_one_rank_reaches_every_rank(oracle):
    """Rank 1's search fails: it still enters the exchange, both ranks report the error, neither
    blocks, and the next exchange on the same group is complete and correct."""
    import torch.multiprocessing as mp
    rng = np.random.default_rng(7)
    t = rng.integers(0, 256, (600, 32), dtype=np.uint8)
    q = t[rng.integers(0, len(t), 40)].copy()
    ridx, rdist = oracle.knn2(q, t)
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, t, out, 1)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0          # joined in time: no rank was left in the collective
    for r in range(2):
        assert out[r] == ("error", -6)
        assert np.array_equal(out[(r, "again")][0], ridx) and np.array_equal(out[(r, "again")][1], rdist)


def test_bench_spawns_ranks_itself():
    """`python bench.py --gpus 2` outside a launcher starts two ranks with torch.distributed.run
    (before it touches torch or HIP), they meet over gloo, and rank 0's JSON line reports both.
    --dry-run-cpu stops short of the hot path, which needs a GPU."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu",
                        "--steps", "3", "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["steps"] == 3
    assert abs(j["max_over_ranks_s"] - 0.002) < 1e-9           # rank 1's figure won the reduction
    assert j["shard_bounds"] == [0, 5_000_000, 10_000_000]
