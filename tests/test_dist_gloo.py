"""gloo tests (CPU, world size 2 and 3) of the sharded search plumbing: shard ranges and id bases, the
entry wire format (16 B: f32 dist, u32 flags, i64 id), the all-gather layout [G, B, kc, 16], the
(distance, id) merge, and ShardedSearcher itself -- the class bench.py runs -- with the oracle standing
in for the two GPU stages (there is no GPU in this container; the GPU stage-1/stage-2 kernels are
covered by tests/test_gpu_parity.py).  Plus bench.py's own launcher under --dry-run."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENTRY = np.dtype([("dist", "<f4"), ("flags", "<u4"), ("id", "<i8")])


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, top_k, ret):
    for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
        sys.path.insert(0, p)
    from coderag_hip.dist import all_gather_entries, shard_range
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows = O.synth_unit_rows(n, d, 11, "f32")
    qs = O.synth_unit_rows(3, d, 12, "f32")
    classes = (np.arange(n) % 3).astype(np.uint8)
    lo, hi = shard_range(n, rank, world)
    kc = 2 * top_k
    ent = np.zeros((3, kc), ENTRY)
    ent["dist"], ent["id"] = np.inf, np.iinfo(np.int64).max
    for b in range(3):
        ids, dd = O.knn_candidates(qs[b], rows[lo:hi], kc, "l2", id_base=lo)
        ent["dist"][b, :len(ids)], ent["id"][b, :len(ids)] = dd, ids
        ent["flags"][b, :len(ids)] = classes[ids]
    local = torch.from_numpy(ent.view(np.uint8).reshape(3, kc, 16).copy())
    g = all_gather_entries(local).numpy().reshape(world, 3, kc * 16).view(ENTRY)   # [G, B, kc]
    ok = True
    for b in range(3):
        allent = np.concatenate([g[r, b] for r in range(world)])
        allent = allent[allent["id"] != np.iinfo(np.int64).max]
        order = np.lexsort((allent["id"], allent["dist"]))[:kc]
        want_ids, want_d = O.knn_candidates(qs[b], rows, kc, "l2")
        ok &= np.array_equal(allent["id"][order], want_ids) and np.array_equal(allent["dist"][order], want_d)
        ok &= np.array_equal(allent["flags"][order], classes[want_ids])
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [1001, 37])
def test_two_rank_gather_and_merge(n):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, 32, 10, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _searcher_worker(rank, world, port, n, d, top_k, sb, ret):
    """ShardedSearcher.search_device end to end over gloo: stage 1 = this shard's exact candidates (oracle) with
    class / symbol flags, the class's own all-gather, stage 2 = (distance, id) merge + the oracle's rerank."""
    for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
        sys.path.insert(0, p)
    from coderag_hip.dist import ShardedSearcher, shard_range
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 4
    rows = O.synth_unit_rows(n, d, 21, "f32")
    qs = O.synth_unit_rows(nq, d, 22, "f32")
    rng = np.random.default_rng(23)
    classes = rng.integers(0, 3, size=n).astype(np.uint8)
    match = (rng.random((nq, n)) < 0.2).astype(np.uint8)
    lo, hi = shard_range(n, rank, world)
    kc = 2 * top_k
    big = np.iinfo(np.int64).max

    def stage1(queries, k, entries, qh, qo):
        ent = np.zeros((nq, kc), ENTRY)
        ent["dist"], ent["id"] = np.inf, big
        for b in range(nq):
            ids, dd = O.knn_candidates(queries[b].numpy(), rows[lo:hi], kc, "l2", id_base=lo)
            ent["dist"][b, :len(ids)], ent["id"][b, :len(ids)] = dd, ids
            ent["flags"][b, :len(ids)] = classes[ids] | (match[b][ids].astype(np.uint32) << 2)
        entries.copy_(torch.from_numpy(ent.view(np.uint8).reshape(nq, kc, 16).copy()))

    def stage2(gathered, G, b, k, boost):
        g = gathered.numpy().reshape(G, b, kc * 16).view(ENTRY)
        out = []
        for i in range(b):
            a = np.concatenate([g[r, i] for r in range(G)])
            a = a[a["id"] != big]
            a = a[np.lexsort((a["id"], a["dist"]))[:kc]]
            sel, sc = O.rerank_numeric(a["dist"], a["flags"] & 3, (a["flags"] >> 2) & 1, k, boost)
            out.append((a["id"][sel], sc, a["dist"][sel]))
        return out

    s = ShardedSearcher(None, rank, world, stage1=stage1, stage2=stage2)
    got = s.search_device(torch.from_numpy(qs), top_k, sb)
    ok = True
    for b in range(nq):
        wi, ws, wd = O.oracle_search_numeric(qs[b], rows, classes, match[b], top_k, sb, "l2")
        ok &= np.array_equal(got[b][0], wi) and np.array_equal(got[b][1], ws) and np.array_equal(got[b][2], wd)
    try:
        ShardedSearcher(None, rank, world + 1)
        ok = False  # a world size that is not the process group's must be refused
    except ValueError:
        pass
    ret[rank] = bool(ok)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n", [(2, 1001), (3, 1000), (3, 37), (3, 2)])
def test_sharded_searcher_class_over_gloo(world, n):
    """Uneven shards (1000 rows over 3 ranks), shards smaller than 2*top_k, and ranks with no rows at all."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_searcher_worker, args=(world, _free_port(), n, 24, 5, 2.0, ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}


@pytest.mark.parametrize("extra", [[], ["--scaling", "strong", "--total-rows", "1001"]])
def test_bench_launches_its_own_ranks_dry_run(extra):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks, rank 0 prints one JSON
    line, exit status 0 (the driver calls bench.py exactly like this on an 8-GPU node)."""
    import json
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--no-extras", "--steps", "2",
                          "--warmup", "1", "--batch", "4"] + extra, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["dry_run"] is True and j["merged_ids_in_range"] is True
    assert j["scaling"] == ("strong" if extra else "weak")
