# -*- coding: utf-8 -*-
"""Row-sharded search over the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

No reference counterpart (the reference has no distributed code); the
partition is the one SURVEY.md section 8(e) derives: rows are independent, so rank r holds the
contiguous slice shard_range(N, r, G) with id_base = its first global row, the query batch is
replicated, every rank computes its exact local 2*top_k (stage 1, crag_search_local_device),
ONE all-gather moves [B, 2*top_k] 16-byte entries per rank (tens of KB: latency-, not
bandwidth-bound, so no bucketing), and every rank merges G lists and reranks (stage 2,
crag_merge_rerank_device).  Exact because the global 2*top_k nearest are a subset of the union
of the shards' 2*top_k nearest.
"""
from __future__ import annotations

from typing import Optional, Tuple

ENTRY_BYTES = 16


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced: the first n_rows % world ranks get one extra row."""
    base, rem = divmod(int(n_rows), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def all_gather_entries(local_entries, group=None):
    """local_entries: uint8 tensor [B, kc, 16] (CUDA for nccl, CPU for gloo).
    Returns [G, B, kc, 16], shard g at index g (ascending id ranges)."""
    import torch
    import torch.distributed as dist
    initialised = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if initialised else 1
    out = torch.empty((world,) + tuple(local_entries.shape), dtype=local_entries.dtype, device=local_entries.device)
    if not initialised:  # a single process without a process group
        out[0].copy_(local_entries)
        return out
    dist.all_gather_into_tensor(out.view(-1), local_entries.contiguous().view(-1), group=group)
    return out


class ShardedSearcher:
    """One rank's view of a row-sharded collection.

    `stage1(queries, top_k, entries, q_hashes, q_hash_off)` fills this shard's [B, 2*top_k, 16] entries and
    `stage2(gathered [G, B, 2*top_k, 16], G, B, top_k, symbol_boost)` merges and reranks; they default to the
    HIP entry points (crag_search_local_device, crag_merge_rerank_device).  The CPU tests and `bench.py --dry-run`
    pass host stand-ins for both, so that the class the bench runs is the class the gloo tests run."""

    def __init__(self, collection, rank: int, world: int, group=None, stage1=None, stage2=None):
        self.col = collection  # GpuCollection holding this rank's slice, id_base already set
        self.rank, self.world, self.group = rank, world, group
        self._entries = None
        self._stage1, self._stage2 = stage1, stage2
        # diagnostics of the exchange step (bench.py): when `timing` is a dict, search_device() waits for the device
        # around the all-gather and adds its seconds to timing["allgather_s"] (and counts timing["calls"]).  Off (None)
        # in the timed region of the bench: the waits would serialise the step.
        self.timing = None
        import torch.distributed as dist
        have = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        if have != world:
            raise ValueError(f"ShardedSearcher: world={world} but the process group has {have} ranks")

    def search_device(self, queries, top_k: int = 10, symbol_boost: float = 2.0, q_hashes=None, q_hash_off=None):
        import torch
        b = queries.shape[0]
        kc = 2 * top_k
        if self._entries is None or tuple(self._entries.shape) != (b, kc, ENTRY_BYTES):
            self._entries = torch.empty((b, kc, ENTRY_BYTES), dtype=torch.uint8, device=queries.device)
        if self._stage1 is not None:
            self._stage1(queries, top_k, self._entries, q_hashes, q_hash_off)
        else:
            self.col.search_local_device(queries, top_k, self._entries, q_hashes, q_hash_off)
        if self.timing is not None and queries.is_cuda:
            import time
            torch.cuda.synchronize(queries.device)
            t0 = time.perf_counter()
            gathered = all_gather_entries(self._entries, self.group)
            torch.cuda.synchronize(queries.device)
            self.timing["allgather_s"] = self.timing.get("allgather_s", 0.0) + time.perf_counter() - t0
            self.timing["calls"] = self.timing.get("calls", 0) + 1
        else:
            gathered = all_gather_entries(self._entries, self.group)
        if self._stage2 is not None:
            return self._stage2(gathered, self.world, b, top_k, symbol_boost)
        from .collection import merge_rerank_device
        return merge_rerank_device(self.col.device, gathered, self.world, b, top_k, symbol_boost)
