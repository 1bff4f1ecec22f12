# -*- coding: utf-8 -*-
"""ctypes front end of oracle/crag_oracle.c  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
`build()` compiles the C file with gcc (no reference sources are involved: nothing on this
path is compiled code in /root/reference, see DESIGN.md "Oracle").
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "crag_oracle.c")
_OUT = os.path.join(_HERE, "_build", "libcrag_oracle.so")
_lib = None

DT = {"f32": 0, "bf16": 1}
METRIC = {"l2": 0, "ip": 1}


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _OUT, "-lm"]
        subprocess.run(cmd, check=True)
    return _OUT


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_OUT):
            build()
        _lib = ctypes.CDLL(_OUT)
        _lib.oracle_knn.restype = ctypes.c_int
        _lib.oracle_distances.restype = ctypes.c_int
        _lib.oracle_rerank.restype = ctypes.c_int
        _lib.oracle_knn_fast_f32.restype = ctypes.c_int
    return _lib


def _dtype_code(rows: np.ndarray) -> int:
    if rows.dtype == np.float32:
        return DT["f32"]
    if rows.dtype == np.uint16:
        return DT["bf16"]
    raise TypeError(rows.dtype)


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def distances(rows: np.ndarray, queries: np.ndarray, metric: str = "l2") -> np.ndarray:
    rows = np.ascontiguousarray(rows)
    queries = np.ascontiguousarray(queries).reshape(-1, rows.shape[1])
    assert queries.dtype == rows.dtype
    out = np.empty((queries.shape[0], rows.shape[0]), np.float32)
    rc = lib().oracle_distances(_p(rows), ctypes.c_int64(rows.shape[0]), ctypes.c_int(rows.shape[1]),
                                ctypes.c_int(_dtype_code(rows)), ctypes.c_int(METRIC[metric]), _p(queries),
                                ctypes.c_int(queries.shape[0]), _p(out))
    assert rc == 0
    return out


def knn(rows: np.ndarray, queries: np.ndarray, k: int, metric: str = "l2",
        id_base: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(ids [B,k] int64, dist [B,k] f32, cnt [B] i32), ordered by (dist, id)."""
    rows = np.ascontiguousarray(rows)
    queries = np.ascontiguousarray(queries).reshape(-1, rows.shape[1])
    assert queries.dtype == rows.dtype
    b = queries.shape[0]
    ids = np.empty((b, k), np.int64)
    dist = np.empty((b, k), np.float32)
    cnt = np.empty(b, np.int32)
    rc = lib().oracle_knn(_p(rows), ctypes.c_int64(rows.shape[0]), ctypes.c_int(rows.shape[1]),
                          ctypes.c_int(_dtype_code(rows)), ctypes.c_int(METRIC[metric]), _p(queries),
                          ctypes.c_int(b), ctypes.c_int(k), ctypes.c_int64(id_base), _p(ids), _p(dist), _p(cnt))
    assert rc == 0
    return ids, dist, cnt


def rerank(dist: np.ndarray, cls: np.ndarray, match: np.ndarray, top_k: int,
           symbol_boost: float) -> Tuple[np.ndarray, np.ndarray]:
    dist = np.ascontiguousarray(dist, np.float32)
    cls = np.ascontiguousarray(cls, np.uint8)
    match = np.ascontiguousarray(match, np.uint8)
    pos = np.empty(max(top_k, 1), np.int32)
    sc = np.empty(max(top_k, 1), np.float64)
    n = lib().oracle_rerank(_p(dist), _p(cls), _p(match), ctypes.c_int(len(dist)), ctypes.c_int(top_k),
                            ctypes.c_double(symbol_boost), _p(pos), _p(sc))
    return pos[:n].astype(np.int64), sc[:n].copy()


def search_numeric(rows, queries, classes, match_rows, top_k, symbol_boost, metric="l2", id_base=0):
    """Full path on numeric columns for a batch.  match_rows: bool [B, N] (or None).
    Returns lists per query of (ids, scores, dists)."""
    ids, dist, cnt = knn(rows, queries, 2 * top_k, metric, 0)
    out = []
    for b in range(ids.shape[0]):
        c = int(cnt[b])
        pos = ids[b, :c]
        m = match_rows[b][pos] if match_rows is not None else np.zeros(c, np.uint8)
        sel, sc = rerank(dist[b, :c], classes[pos], m, top_k, symbol_boost)
        out.append((pos[sel] + id_base, sc, dist[b, :c][sel]))
    return out


def knn_fast_f32(rows: np.ndarray, queries: np.ndarray, k: int):
    rows = np.ascontiguousarray(rows)
    queries = np.ascontiguousarray(queries).reshape(-1, rows.shape[1])
    b = queries.shape[0]
    ids = np.empty((b, k), np.int64)
    dist = np.empty((b, k), np.float32)
    rc = lib().oracle_knn_fast_f32(_p(rows), ctypes.c_int64(rows.shape[0]), ctypes.c_int(rows.shape[1]),
                                   ctypes.c_int(_dtype_code(rows)), _p(queries), ctypes.c_int(b),
                                   ctypes.c_int(k), _p(ids), _p(dist))
    assert rc == 0
    return ids, dist
