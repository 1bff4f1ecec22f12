# -*- coding: utf-8 -*-
"""GpuCollection: one code_chunks__<workspace> embedding collection resident in HBM.

Stands where a chromadb.Collection stands in the reference
(/root/reference/retriever/hybrid_search.py:28-39, indexer/embed_ingest.py:38-54,93):
`add` takes embeddings plus the two metadata fields the scorer reads (name, path --
hybrid_search.py:98-99), `search` does query + rerank on the GPU through the C ABI.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import CragError, check

PATH_NEUTRAL, PATH_PENALISED, PATH_BOOSTED = 0, 1, 2


def f32_to_bf16_bits(a: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns (uint16), round-to-nearest-even (same rule as the C side)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32)
    out = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    nan = np.isnan(a)
    if nan.any():
        out = np.where(nan, ((u >> np.uint32(16)) | np.uint32(0x40)).astype(np.uint16), out)
    return out


def classify_path(path) -> int:
    """Per-chunk class of the if/elif chain at hybrid_search.py:99-111 (query independent,
    so it is computed once at ingest and stored as one byte per row)."""
    p = str(path).lower()
    if "/test/" in p or "/tests/" in p or "/spec/" in p or "/__tests__/" in p:
        return PATH_PENALISED
    if p.endswith(".d.ts"):
        return PATH_PENALISED
    if "/src/" in p or "/lib/" in p:
        return PATH_BOOSTED
    return PATH_NEUTRAL


def lowered_name_bytes(name) -> bytes:
    """UTF-8 of str(name).lower() -- the string hybrid_search.py:98 tests with `in q_lower`.
    UTF-8 is self-synchronising, so byte-substring == code-point-substring."""
    return str(name).lower().encode("utf-8", "surrogatepass")


def hash_names(names: Sequence) -> Tuple[np.ndarray, np.ndarray]:
    """(uint64 FNV-1a hashes, uint16 byte lengths) of the lowered names; empty name -> (0, 0)."""
    lib = _lib.load()
    bs = [lowered_name_bytes(n) for n in names]
    off = np.zeros(len(bs) + 1, np.int64)
    np.cumsum([len(b) for b in bs], out=off[1:])
    buf = b"".join(bs)
    out = np.zeros(len(bs), np.uint64)
    if len(bs):
        lib.crag_hash_names(buf, off.ctypes.data, len(bs), out.ctypes.data)
    lens = np.minimum(np.diff(off), 65535).astype(np.uint16)
    return out, lens


def _ptr(a) -> Optional[int]:
    return None if a is None else a.ctypes.data


class GpuCollection:
    def __init__(self, dim: int, dtype: str = "f32", metric: str = "l2", device: int = 0):
        self._lib = _lib.load()
        self.dim = int(dim)
        self.dtype = dtype
        self.metric = metric
        self.device = int(device)
        h = C.c_void_p()
        check(self._lib.crag_index_create(self.dim, {"f32": 0, "bf16": 1}[dtype], {"l2": 0, "ip": 1}[metric],
                                          self.device, C.byref(h)), "crag_index_create")
        self._h = h
        self._keep = []  # tensors adopted zero-copy must outlive the handle
        self._finalized = False

    # -- lifetime ---------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.crag_index_destroy(self._h)
            self._h = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __len__(self) -> int:
        return int(self._lib.crag_index_size(self._h))

    @property
    def handle(self) -> C.c_void_p:
        return self._h

    @property
    def dim_pad(self) -> int:
        return int(self._lib.crag_index_dim_pad(self._h))

    def set_id_base(self, base: int):
        check(self._lib.crag_index_set_id_base(self._h, int(base)))

    # -- ingest -----------------------------------------------------------------------
    def _rows_host(self, rows: np.ndarray) -> np.ndarray:
        rows = np.asarray(rows)
        if rows.ndim != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"rows must be [n, {self.dim}], got {rows.shape}")
        if self.dtype == "f32":
            return np.ascontiguousarray(rows, np.float32)
        if rows.dtype == np.uint16:
            return np.ascontiguousarray(rows)
        return f32_to_bf16_bits(rows)

    def add(self, rows: np.ndarray, names: Optional[Sequence] = None, paths: Optional[Sequence] = None,
            name_hash: Optional[np.ndarray] = None, name_len: Optional[np.ndarray] = None,
            path_class: Optional[np.ndarray] = None):
        """Host ingest (the col.add of embed_ingest.py:93, embeddings supplied by the caller)."""
        r = self._rows_host(rows)
        n = r.shape[0]
        if names is not None:
            name_hash, name_len = hash_names(names)
        if paths is not None:
            path_class = np.fromiter((classify_path(p) for p in paths), np.uint8, count=n)
        for a, dt in ((name_hash, np.uint64), (name_len, np.uint16), (path_class, np.uint8)):
            if a is not None and (a.dtype != dt or a.shape != (n,)):
                raise ValueError("metadata columns must be 1-D of length n with the documented dtypes")
        nh = None if name_hash is None else np.ascontiguousarray(name_hash)
        nl = None if name_len is None else np.ascontiguousarray(name_len)
        pc = None if path_class is None else np.ascontiguousarray(path_class)
        check(self._lib.crag_index_add(self._h, r.ctypes.data, n, _ptr(nh), _ptr(nl), _ptr(pc)), "crag_index_add")
        self._finalized = False

    def _check_tensor(self, t, n=None):
        import torch
        want = torch.float32 if self.dtype == "f32" else torch.bfloat16
        if t.dtype != want or not t.is_cuda or not t.is_contiguous() or t.dim() != 2 or t.shape[1] != self.dim:
            raise ValueError(f"need a contiguous CUDA {want} tensor [n, {self.dim}]")
        if t.device.index != self.device:
            raise ValueError("tensor is on another device")

    def add_device(self, rows, name_hash=None, path_class=None):
        """Device ingest: rows is a CUDA torch tensor in the index dtype; columns are CUDA tensors
        (int64/uint64 bit patterns for name_hash, uint8 for path_class)."""
        import torch
        self._check_tensor(rows)
        torch.cuda.current_stream(self.device).synchronize()
        check(self._lib.crag_index_add_device(self._h, rows.data_ptr(), rows.shape[0],
                                              None if name_hash is None else name_hash.data_ptr(),
                                              None if path_class is None else path_class.data_ptr()),
              "crag_index_add_device")
        self._finalized = False

    def attach_device(self, rows, name_hash=None, path_class=None):
        """Zero-copy: adopt the tensors' memory as the whole collection."""
        import torch
        self._check_tensor(rows)
        torch.cuda.current_stream(self.device).synchronize()
        check(self._lib.crag_index_attach_device(self._h, rows.data_ptr(), rows.shape[0],
                                                 None if name_hash is None else name_hash.data_ptr(),
                                                 None if path_class is None else path_class.data_ptr()),
              "crag_index_attach_device")
        self._keep = [rows, name_hash, path_class]
        self._finalized = False

    def rows_host(self, first: int = 0, n: Optional[int] = None) -> np.ndarray:
        """Rows [first, first + n) copied back from the device: float32 [n, dim], or uint16 bf16 bit patterns."""
        n = len(self) - first if n is None else int(n)
        out = np.empty((n, self.dim), np.float32 if self.dtype == "f32" else np.uint16)
        check(self._lib.crag_index_get_rows(self._h, int(first), n, out.ctypes.data), "crag_index_get_rows")
        return out

    def set_symbols(self, hashes: np.ndarray, lens: np.ndarray):
        hashes = np.ascontiguousarray(hashes, np.uint64)
        lens = np.ascontiguousarray(lens, np.uint16)
        check(self._lib.crag_index_set_symbols(self._h, hashes.ctypes.data, lens.ctypes.data, len(hashes)))

    def finalize(self):
        check(self._lib.crag_index_finalize(self._h), "crag_index_finalize")
        self._finalized = True

    # -- search -----------------------------------------------------------------------
    @staticmethod
    def _qtext(query_texts: Optional[Sequence[str]], b: int):
        if query_texts is None:
            return None, None, None
        if len(query_texts) != b:
            raise ValueError("one query text per query vector")
        bs = [str(q).lower().encode("utf-8", "surrogatepass") for q in query_texts]  # q_lower, :93
        off = np.zeros(b + 1, np.int64)
        np.cumsum([len(x) for x in bs], out=off[1:])
        return b"".join(bs) + b"\0", off, bs

    def search(self, queries: np.ndarray, top_k: int = 10, symbol_boost: float = 2.0,
               query_texts: Optional[Sequence[str]] = None):
        """Batch form of HybridSearcher.search on host buffers.
        Returns (ids int64 [B,top_k], scores f64, dists f32, counts i32)."""
        if not self._finalized:
            self.finalize()
        q = np.ascontiguousarray(np.asarray(queries, np.float32).reshape(-1, self.dim))
        b = q.shape[0]
        ids = np.empty((b, top_k), np.int64)
        sc = np.empty((b, top_k), np.float64)
        di = np.empty((b, top_k), np.float32)
        cn = np.empty(b, np.int32)
        buf, off, _ = self._qtext(query_texts, b)
        check(self._lib.crag_search(self._h, q.ctypes.data, b, int(top_k), float(symbol_boost), buf, _ptr(off),
                                    ids.ctypes.data, sc.ctypes.data, di.ctypes.data, cn.ctypes.data), "crag_search")
        return ids, sc, di, cn

    def symbol_hashes(self, query_text: str, cap: int = 256) -> np.ndarray:
        """Sorted hashes of the chunk names that occur in the lowered query text -- ALL of them: the buffer grows
        until the library's answer fits (the search takes this list as the query's symbol side)."""
        qb = str(query_text).lower().encode("utf-8", "surrogatepass")
        while True:
            out = np.zeros(cap, np.uint64)
            n = check(self._lib.crag_symbol_hashes(self._h, qb, len(qb), out.ctypes.data, cap), "crag_symbol_hashes")
            if n < cap:
                return out[:n]
            cap *= 8

    def search_device(self, queries, top_k: int = 10, symbol_boost: float = 2.0, q_hashes=None, q_hash_off=None,
                      out=None, stream=None):
        """queries: CUDA tensor [B, dim] in the index dtype.  Returns CUDA tensors
        (ids int64, scores float64, dists float32, counts int32)."""
        import torch
        if not self._finalized:
            self.finalize()
        self._check_tensor(queries)
        b = queries.shape[0]
        dev = queries.device
        if out is None:
            out = (torch.empty((b, top_k), dtype=torch.int64, device=dev),
                   torch.empty((b, top_k), dtype=torch.float64, device=dev),
                   torch.empty((b, top_k), dtype=torch.float32, device=dev),
                   torch.empty((b,), dtype=torch.int32, device=dev))
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        check(self._lib.crag_search_device(self._h, queries.data_ptr(), b, int(top_k), float(symbol_boost),
                                           None if q_hashes is None else q_hashes.data_ptr(),
                                           None if q_hash_off is None else q_hash_off.data_ptr(),
                                           out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                           out[3].data_ptr(), st), "crag_search_device")
        return out

    def search_local_device(self, queries, top_k: int, entries, q_hashes=None, q_hash_off=None, stream=None):
        """Stage 1 of the sharded search: this shard's exact 2*top_k nearest per query into
        `entries` (CUDA uint8 tensor [B, 2*top_k, 16])."""
        import torch
        if not self._finalized:
            self.finalize()
        self._check_tensor(queries)
        st = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
        check(self._lib.crag_search_local_device(self._h, queries.data_ptr(), queries.shape[0], int(top_k),
                                                 None if q_hashes is None else q_hashes.data_ptr(),
                                                 None if q_hash_off is None else q_hash_off.data_ptr(),
                                                 entries.data_ptr(), st), "crag_search_local_device")
        return entries

    def stats(self, reset: bool = False) -> dict:
        a = np.zeros(9, np.int64)
        check(self._lib.crag_stats(self._h, a.ctypes.data, int(reset)))
        keys = ["searches", "queries", "fallback", "uncertified", "overflow", "mfma_batches", "gemv_batches", "exact_scan", "rescan"]
        return dict(zip(keys, (int(x) for x in a)))

    def debug_dense_scores(self, queries, path: str = "gemv"):
        """Test hook: approximate scan scores of every row, CUDA float32 [B, rows]."""
        import torch
        if not self._finalized:
            self.finalize()
        self._check_tensor(queries)
        out = torch.empty((queries.shape[0], len(self)), dtype=torch.float32, device=queries.device)
        st = torch.cuda.current_stream(self.device).cuda_stream
        check(self._lib.crag_debug_dense_scores(self._h, queries.data_ptr(), queries.shape[0],
                                                {"gemv": 1, "mfma": 2, "tiled": 3}[path], out.data_ptr(), st),
              "crag_debug_dense_scores")
        return out

    def set_profiling(self, on: bool):
        check(self._lib.crag_set_profiling(self._h, int(bool(on))))

    def profile(self, reset: bool = False) -> dict:
        a = np.zeros(4, np.float64)
        check(self._lib.crag_get_profile(self._h, a.ctypes.data, int(reset)))
        return {"sample_scan_ms": float(a[0]), "filter_scan_ms": float(a[1]), "searches": int(a[2]),
                "filter_scan_rows": float(a[3])}

    def set_path(self, path: str):
        check(self._lib.crag_set_path(self._h, {"auto": 0, "gemv": 1, "mfma": 2, "tiled": 3}[path]))


def merge_rerank_device(device: int, entries, n_shards: int, b: int, top_k: int, symbol_boost: float, out=None,
                        stream=None):
    """Stage 2 of the sharded search on CUDA tensors: entries [G, B, 2*top_k, 16] uint8."""
    import torch
    lib = _lib.load()
    dev = entries.device
    if out is None:
        out = (torch.empty((b, top_k), dtype=torch.int64, device=dev),
               torch.empty((b, top_k), dtype=torch.float64, device=dev),
               torch.empty((b, top_k), dtype=torch.float32, device=dev),
               torch.empty((b,), dtype=torch.int32, device=dev))
    st = torch.cuda.current_stream(device).cuda_stream if stream is None else stream
    check(lib.crag_merge_rerank_device(int(device), entries.data_ptr(), int(n_shards), int(b), int(top_k),
                                       float(symbol_boost), out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(),
                                       out[3].data_ptr(), st), "crag_merge_rerank_device")
    return out


def search_multi(collections: List[GpuCollection], queries: np.ndarray, top_k: int = 10, symbol_boost: float = 2.0,
                 query_texts: Optional[Sequence[str]] = None, query_hashes: Optional[Sequence[np.ndarray]] = None):
    """One query per collection handle (workspace isolation), one grouped launch.  The symbol boost takes either the
    query texts (hashed by the library) or `query_hashes`: per query the array GpuCollection.symbol_hashes() returned,
    for a caller that computed them already (HybridSearcher does, to verify them against the names as strings)."""
    lib = _lib.load()
    b = len(collections)
    dim = collections[0].dim
    for c in collections:
        if not c._finalized:
            c.finalize()
    q = np.ascontiguousarray(np.asarray(queries, np.float32).reshape(b, dim))
    hs = (C.c_void_p * b)(*[c.handle for c in collections])
    ids = np.empty((b, top_k), np.int64)
    sc = np.empty((b, top_k), np.float64)
    di = np.empty((b, top_k), np.float32)
    cn = np.empty(b, np.int32)
    if query_hashes is not None:
        if len(query_hashes) != b:
            raise ValueError("one hash array per query vector")
        if b == 1:   # the /search case: no concatenation
            hoff = np.array([0, len(query_hashes[0])], np.int32)
            cat = np.ascontiguousarray(query_hashes[0], np.uint64) if hoff[1] else np.zeros(1, np.uint64)
        else:
            hoff = np.zeros(b + 1, np.int32)
            np.cumsum([len(h) for h in query_hashes], out=hoff[1:])
            cat = np.ascontiguousarray(np.concatenate(query_hashes), np.uint64) if hoff[-1] else np.zeros(1, np.uint64)
        check(lib.crag_search_multi_hashed(hs, q.ctypes.data, b, int(top_k), float(symbol_boost), cat.ctypes.data,
                                           hoff.ctypes.data, ids.ctypes.data, sc.ctypes.data, di.ctypes.data,
                                           cn.ctypes.data), "crag_search_multi_hashed")
        return ids, sc, di, cn
    buf, off, _ = GpuCollection._qtext(query_texts, b)
    check(lib.crag_search_multi(hs, q.ctypes.data, b, int(top_k), float(symbol_boost), buf, _ptr(off),
                                ids.ctypes.data, sc.ctypes.data, di.ctypes.data, cn.ctypes.data), "crag_search_multi")
    return ids, sc, di, cn


def search_multi_device(collections: List[GpuCollection], queries, top_k: int = 10, symbol_boost: float = 2.0,
                        q_hashes=None, q_hash_off=None, out=None, stream=None):
    """One query per collection handle, everything on the device: queries is a CUDA tensor [B, dim] in the
    index dtype, q_hashes / q_hash_off (sorted symbol hashes per query, B+1 int32 offsets) may be None.
    Returns CUDA tensors (ids int64, scores float64, dists float32, counts int32)."""
    import torch
    lib = _lib.load()
    b = len(collections)
    for c in collections:
        if not c._finalized:
            c.finalize()
    collections[0]._check_tensor(queries)
    dev = queries.device
    hs = (C.c_void_p * b)(*[c.handle for c in collections])
    if out is None:
        out = (torch.empty((b, top_k), dtype=torch.int64, device=dev),
               torch.empty((b, top_k), dtype=torch.float64, device=dev),
               torch.empty((b, top_k), dtype=torch.float32, device=dev),
               torch.empty((b,), dtype=torch.int32, device=dev))
    st = torch.cuda.current_stream(collections[0].device).cuda_stream if stream is None else stream
    check(lib.crag_search_multi_device(hs, queries.data_ptr(), b, int(top_k), float(symbol_boost),
                                       None if q_hashes is None else q_hashes.data_ptr(),
                                       None if q_hash_off is None else q_hash_off.data_ptr(),
                                       out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), out[3].data_ptr(), st),
          "crag_search_multi_device")
    return out
