# -*- coding: utf-8 -*-
"""ctypes front end of oracle/hnsw_oracle.c  --  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT CODE.

A restatement of hnswlib's published algorithm with Chroma's documented defaults (l2, M = 16, ef_construction = 100,
ef_search = 100): the approximate index behind `col.query()` at /root/reference/retriever/hybrid_search.py:60-64.
PARITY UNPINNED (see the C file's header).  Only tests/ and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "hnsw_oracle.c")
_OUT = os.path.join(_HERE, "_build", "libhnsw_oracle.so")
_lib = None

CHROMA_DEFAULTS = {"M": 16, "ef_construction": 100, "ef_search": 100}


def build(force: bool = False) -> str:
    os.makedirs(os.path.dirname(_OUT), exist_ok=True)
    if force or not os.path.exists(_OUT) or os.path.getmtime(_OUT) < os.path.getmtime(_SRC):
        # AVX2 + FMA, not -march=native: the library built in the build container travels to the GPU box's host
        subprocess.run(["gcc", "-O3", "-mavx2", "-mfma", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _OUT, "-lm"], check=True)
    return _OUT


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_OUT)
        _lib.hnsw_build.restype = ctypes.c_void_p
        _lib.hnsw_build.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        _lib.hnsw_search.restype = ctypes.c_int
        _lib.hnsw_search.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p]
        _lib.hnsw_stats.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
        _lib.hnsw_free.argtypes = [ctypes.c_void_p]
    return _lib


class HnswIndex:
    """index = HnswIndex(rows_f32); ids, dist = index.query(queries_f32, k)"""

    def __init__(self, rows: np.ndarray, M: int = 16, ef_construction: int = 100, seed: int = 100, threads: int = 1):
        self.rows = np.ascontiguousarray(rows, dtype=np.float32)  # the index borrows this memory
        n, d = self.rows.shape
        self._h = lib().hnsw_build(self.rows.ctypes.data, n, d, M, ef_construction, seed, threads)
        if not self._h:
            raise ValueError("hnsw_build: empty input or M < 2")

    def query(self, queries: np.ndarray, k: int, ef_search: int = 100, threads: int = 1):
        q = np.ascontiguousarray(queries, dtype=np.float32)
        assert q.shape[1] == self.rows.shape[1]
        ids = np.empty((q.shape[0], k), np.int64)
        dist = np.empty((q.shape[0], k), np.float32)
        rc = lib().hnsw_search(self._h, q.ctypes.data, q.shape[0], k, ef_search, threads, ids.ctypes.data, dist.ctypes.data)
        if rc:
            raise RuntimeError("hnsw_search failed")
        return ids, dist

    def stats(self):
        deg, top = ctypes.c_double(), ctypes.c_int()
        lib().hnsw_stats(self._h, ctypes.byref(deg), ctypes.byref(top))
        return {"mean_degree_level0": deg.value, "max_level": top.value}

    def close(self):
        if self._h:
            lib().hnsw_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
