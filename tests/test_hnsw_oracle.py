"""oracle/hnsw_oracle.c -- the HNSW restatement behind bench.py's approximate-index CPU leg (Chroma's defaults: l2, M = 16,
ef_construction = 100, ef_search = 100; the step at /root/reference/retriever/hybrid_search.py:60-64).  PARITY UNPINNED:
neither chromadb nor chroma-hnswlib exists here and the reference holds no fixture for this step, so these tests check
the restatement against the exact neighbours and against the structural rules of the published algorithm, nothing more.
"""
import numpy as np
import pytest

from oracle import hnsw


def _unit(rng, n, d):
    x = rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


def _exact(x, q, k):
    d2 = ((q[:, None, :] - x[None, :, :]) ** 2).sum(-1) if x.shape[0] * q.shape[0] * x.shape[1] < 5e7 else \
        (x * x).sum(1)[None, :] - 2.0 * (q @ x.T) + (q * q).sum(1)[:, None]
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return order, np.take_along_axis(d2, order, axis=1)


def _recall(ids, exact):
    return float(np.mean([len(set(a.tolist()) & set(b.tolist())) / exact.shape[1] for a, b in zip(ids, exact)]))


@pytest.mark.parametrize("threads", [1, 4])
def test_low_dimensional_rows_are_found_almost_exactly(threads):
    rng = np.random.default_rng(3)
    x, q = _unit(rng, 4000, 8), _unit(rng, 64, 8)
    ix = hnsw.HnswIndex(x, threads=threads)
    ids, dist = ix.query(q, 20, threads=threads)
    exact, exact_d = _exact(x, q, 20)
    assert _recall(ids, exact) >= 0.97
    # what it returns is sorted by (distance, id), unique, in range, and carries the true squared-L2 distances
    assert (np.diff(dist, axis=1) >= 0).all()
    assert all(len(set(r.tolist())) == 20 for r in ids) and ids.min() >= 0 and ids.max() < len(x)
    true_d = ((q[:, None, :] - x[ids]) ** 2).sum(-1)
    np.testing.assert_allclose(dist, true_d, rtol=1e-4, atol=1e-6)
    st = ix.stats()
    assert 0 < st["mean_degree_level0"] <= 32 and 1 <= st["max_level"] <= 6  # log_16(4000) ~ 3
    ix.close()


def test_clustered_rows_high_recall_isotropic_rows_low_recall():
    """A graph index lives on structure: embeddings that cluster (low intrinsic dimension) are found, isotropic rows
    in many dimensions are not -- the reason the bench's synthetic corpus understates the reference index's recall."""
    rng = np.random.default_rng(5)
    centres = _unit(rng, 40, 96)
    x = centres[rng.integers(0, 40, 6000)] + 0.05 * rng.standard_normal((6000, 96)).astype(np.float32)
    x = (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
    q = x[rng.integers(0, 6000, 50)] + 0.01 * rng.standard_normal((50, 96)).astype(np.float32)
    ix = hnsw.HnswIndex(x, threads=4)
    rec_clustered = _recall(ix.query(q, 10, threads=4)[0], _exact(x, q.astype(np.float32), 10)[0])
    ix.close()
    y, qy = _unit(rng, 6000, 96), _unit(rng, 50, 96)
    iy = hnsw.HnswIndex(y, threads=4)
    rec_iso = _recall(iy.query(qy, 10, threads=4)[0], _exact(y, qy, 10)[0])
    rec_iso_wide = _recall(iy.query(qy, 10, ef_search=3000, threads=4)[0], _exact(y, qy, 10)[0])
    iy.close()
    assert rec_clustered >= 0.9
    assert rec_iso < rec_clustered
    assert rec_iso_wide >= 0.97  # a wide enough beam reaches everything: the graph is connected


def test_single_thread_build_is_deterministic_and_edge_sizes_work():
    rng = np.random.default_rng(7)
    x, q = _unit(rng, 1500, 16), _unit(rng, 8, 16)
    a = hnsw.HnswIndex(x, threads=1, seed=11)
    b = hnsw.HnswIndex(x, threads=1, seed=11)
    ia, da = a.query(q, 10)
    ib, db = b.query(q, 10)
    assert (ia == ib).all() and (da == db).all()
    a.close()
    b.close()
    # fewer rows than k: the tail is padded with -1 / inf; one row; k = 1
    tiny = hnsw.HnswIndex(x[:5], threads=1)
    ids, dist = tiny.query(q[:2], 10)
    assert (np.sort(ids[:, :5], axis=1) == np.arange(5)).all() and (ids[:, 5:] == -1).all() and np.isinf(dist[:, 5:]).all()
    tiny.close()
    one = hnsw.HnswIndex(x[:1], threads=1)
    ids, dist = one.query(q[:1], 1)
    assert ids[0, 0] == 0
    one.close()
    with pytest.raises(ValueError):
        hnsw.HnswIndex(np.zeros((0, 16), np.float32))
