#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the oracle (oracle/oracle.py, numpy statement).

The reference cannot run here (chromadb is not installed) and holds no
golden vectors for this path, so these fixtures pin the ORACLE's behaviour (and through it the
GPU path's), not Chroma's: "parity unpinned" at the Chroma boundary.  Inputs are stored in the
files, so nothing depends on numpy's RNG stream.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, n, d, dtype, metric, n_queries, top_k, symbol_boost
    ("c2_f32_384", 512, 384, "f32", "l2", 3, 10, 2.0),
    ("c3_bf16_768", 384, 768, "bf16", "l2", 4, 100, 2.0),
    ("c5_bf16_1024", 256, 1024, "bf16", "l2", 3, 8, 2.5),
    ("ip_f32_100", 300, 100, "f32", "ip", 2, 5, 2.0),
]


def main():
    for name, n, d, dtype, metric, nq, top_k, sb in CASES:
        seed = 1234
        rows = O.synth_unit_rows(n, d, seed, dtype)
        queries = O.synth_unit_rows(nq, d, seed + 1, dtype)
        rng = np.random.default_rng(seed + 2)
        classes = rng.choice(np.array([0, 1, 2], np.uint8), size=n, p=[0.70, 0.05, 0.25]).astype(np.uint8)
        match = (rng.random((nq, n)) < 0.1).astype(np.uint8)
        ids = np.full((nq, top_k), -1, np.int64)
        scores = np.zeros((nq, top_k), np.float64)
        dists = np.full((nq, top_k), np.inf, np.float32)
        counts = np.zeros(nq, np.int32)
        cand_ids = np.zeros((nq, min(2 * top_k, n)), np.int64)
        cand_d = np.zeros((nq, min(2 * top_k, n)), np.float32)
        for b in range(nq):
            ci, cd = O.knn_candidates(queries[b], rows, 2 * top_k, metric)
            cand_ids[b], cand_d[b] = ci, cd
            i, s, dd = O.oracle_search_numeric(queries[b], rows, classes, match[b], top_k, sb, metric)
            counts[b] = len(i)
            ids[b, :len(i)], scores[b, :len(i)], dists[b, :len(i)] = i, s, dd
        np.savez_compressed(os.path.join(HERE, name + ".npz"), rows=rows, queries=queries, classes=classes,
                            match=match, ids=ids, scores=scores, dists=dists, counts=counts, cand_ids=cand_ids,
                            cand_d=cand_d, top_k=top_k, symbol_boost=sb, metric=metric, dtype=dtype)
        print(name, "ok", counts)


if __name__ == "__main__":
    main()
