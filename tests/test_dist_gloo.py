"""world_size-2 gloo test (CPU) of the sharded search plumbing: shard ranges and id bases, the
entry wire format (16 B: f32 dist, u32 flags, i64 id), the all-gather layout [G, B, kc, 16] and
the (distance, id) merge.  Stage 1 on each rank is played by the oracle here (there is no GPU in
this container); the GPU stage-1/stage-2 kernels are covered by tests/test_gpu_parity.py."""
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
