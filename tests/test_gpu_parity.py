"""GPU parity tests: the HIP path, called through the C ABI (include/coderag_hip.h), against the
CPU oracle on the same seeded inputs.  Bar: ids, float32 distances and float64 scores are
BIT-EXACT (integer/index work and canonical fp64 arithmetic), counts identical.

Run on the GPU box:  python -u -m pytest tests -m gpu -x -v --timeout 120 --timeout-method thread
"""
import glob
import os
import threading

import numpy as np
import pytest

import coderag_hip as ch
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
ENTRY = np.dtype([("dist", "<f4"), ("flags", "<u4"), ("id", "<i8")])


# ---------------------------------------------------------------------------------------------
def names_for_match(match: np.ndarray):
    """Realise an arbitrary bool matrix match[nq, n] through real names and query texts: row i is
    named after its match pattern, query b's text contains exactly the names with bit b set."""
    nq, n = match.shape
    assert nq <= 6
    pat = np.zeros(n, np.int64)
    for b in range(nq):
        pat |= match[b].astype(np.int64) << b
    names = ["" if p == 0 else f"Sym{int(p):02d}q" for p in pat]
    texts = []
    for b in range(nq):
        used = sorted({names[i].lower() for i in range(n) if (pat[i] >> b) & 1})
        texts.append("find " + " and ".join(used) + " please")
    return names, texts


def to_f32(rows):
    return O.bf16_bits_to_f32(rows) if rows.dtype == np.uint16 else rows


def make_collection(rows, dtype, metric="l2", names=None, classes=None, id_base=0):
    col = ch.GpuCollection(rows.shape[1], dtype, metric)
    kw = {}
    if names is not None:
        kw["names"] = names
    if classes is not None:
        kw["path_class"] = np.ascontiguousarray(classes, np.uint8)
    col.add(rows, **kw)
    if id_base:
        col.set_id_base(id_base)
    col.finalize()
    return col


def check_against_oracle(coracle, col, rows, queries, classes, match, top_k, sb, metric, texts, id_base=0):
    ids, sc, di, cn = col.search(to_f32(queries), top_k, sb, query_texts=texts)
    want = coracle.search_numeric(rows, queries, classes, match, top_k, sb, metric, id_base)
    for b, (wi, ws, wd) in enumerate(want):
        n = len(wi)
        assert int(cn[b]) == n, (b, int(cn[b]), n)
        assert np.array_equal(ids[b, :n], wi), (b, ids[b, :n], wi)
        assert np.array_equal(di[b, :n].view(np.uint32), wd.view(np.uint32)), b
        assert np.array_equal(sc[b, :n], ws), b
        assert np.all(ids[b, n:] == -1)


def synth_case(n, d, dtype, nq, seed=1234):
    rows = O.synth_unit_rows(n, d, seed, dtype)
    queries = O.synth_unit_rows(nq, d, seed + 1, dtype)
    rng = np.random.default_rng(seed + 2)
    classes = rng.choice(np.array([0, 1, 2], np.uint8), size=n, p=[0.70, 0.05, 0.25]).astype(np.uint8)
    return rows, queries, classes, rng


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_golden_fixtures(path):
    g = np.load(path, allow_pickle=False)
    rows, queries, match = g["rows"], g["queries"], g["match"]
    top_k, sb, metric, dtype = int(g["top_k"]), float(g["symbol_boost"]), str(g["metric"]), str(g["dtype"])
    names, texts = names_for_match(match)
    col = make_collection(rows, dtype, metric, names, g["classes"])
    ids, sc, di, cn = col.search(to_f32(queries), top_k, sb, query_texts=texts)
    assert np.array_equal(cn, g["counts"])
    for b in range(len(cn)):
        n = int(cn[b])
        assert np.array_equal(ids[b, :n], g["ids"][b, :n])
        assert np.array_equal(sc[b, :n], g["scores"][b, :n])
        assert np.array_equal(di[b, :n].view(np.uint32), g["dists"][b, :n].view(np.uint32))
    assert col.stats()["fallback"] == 0


@pytest.mark.parametrize("dtype,d,n,nq,top_k,metric", [
    ("f32", 384, 20000, 1, 10, "l2"),       # C2 shape, GEMV NQ=1
    ("f32", 384, 20000, 3, 10, "l2"),       # NQ=4 template, 3 real queries
    ("f32", 384, 5000, 8, 8, "ip"),         # NQ=8
    ("bf16", 768, 30000, 1, 10, "l2"),      # C3/C4 corpus, batch 1
    ("bf16", 768, 30000, 5, 100, "l2"),     # top-100 -> 200 candidates, M=250
    ("bf16", 1024, 9000, 2, 8, "l2"),       # C5 shape
    ("bf16", 384, 7000, 4, 10, "ip"),
    ("f32", 768, 3000, 2, 10, "l2"),        # C = 192
    ("f32", 1024, 3000, 1, 5, "l2"),        # C = 256
    ("bf16", 512, 4000, 6, 10, "l2"),       # C = 64
    ("bf16", 256, 4000, 1, 10, "l2"),       # C = 32
    ("f32", 100, 3000, 2, 10, "l2"),        # padded rows (400 B -> 512 B)
    ("f32", 7, 500, 1, 3, "l2"),            # tiny dim
    ("bf16", 1000, 2500, 3, 10, "l2"),      # padded bf16 rows
    ("f32", 384, 20000, 20, 10, "l2"),      # f32, B > 8: several GEMV groups
    ("f32", 384, 20000, 2, 200, "l2"),      # CRAG_MAX_TOP_K through the GEMV
    ("bf16", 768, 600, 1, 200, "l2"),       # top_k close to the collection size
])
def test_parity_gemv_regime(coracle, dtype, d, n, nq, top_k, metric):
    rows, queries, classes, rng = synth_case(n, d, dtype, nq)
    m = min(nq, 6)
    match = np.zeros((nq, n), np.uint8)
    match[:m] = rng.random((m, n)) < 0.08
    names, texts = names_for_match(match[:m])
    texts = texts + ["no symbols here"] * (nq - m)
    col = make_collection(rows, dtype, metric, names, classes)
    col.set_path("gemv")
    check_against_oracle(coracle, col, rows, queries, classes, match, top_k, 2.0, metric, texts)
    assert col.stats()["gemv_batches"] >= 1


@pytest.mark.parametrize("d,n,nq,top_k,metric", [
    (768, 30000, 16, 10, "l2"),
    (768, 30011, 130, 10, "l2"),            # ragged rows and queries (two query tiles)
    (768, 9000, 256, 100, "l2"),            # C3-like: top-100
    (1024, 5000, 33, 8, "l2"),
    (384, 8000, 64, 10, "ip"),
    (256, 300, 17, 10, "l2"),               # fewer rows than one sample tile
    (512, 6000, 200, 10, "l2"),             # B > 128: two query blocks per wave
    (384, 8000, 150, 10, "ip"),
    (256, 3001, 257, 10, "l2"),             # second query group holds one query
    (768, 20000, 300, 10, "ip"),
    (1024, 30011, 200, 10, "ip"),           # 1024-element rows: split-K scan, ragged last tile, two query groups
    (1000, 9000, 17, 10, "l2"),             # padded to 1024; one partly filled wave
    (1024, 40, 9, 10, "l2"),                # a single ragged tile
    (768, 20000, 130, 200, "l2"),           # CRAG_MAX_TOP_K: 400 candidates, M = 500, deepest cascade
    (384, 20000, 20, 200, "ip"),
    (768, 20000, 2, 10, "l2"),              # two queries: one partly filled wave of the two-block cascade
    (384, 7000, 2, 10, "ip"),
    (512, 5000, 2, 100, "l2"),              # two queries, M > 64: the two-block cascade
    (768, 30011, 3, 10, "l2"),              # three queries: two-block cascade, one partly filled wave
    (384, 9000, 260, 10, "l2"),             # more than 256 queries on 384- and 512-element rows: the variants with the
    (512, 7000, 300, 10, "ip"),             # default load policy and the pacing poll, which 256 / 768 above also take
    # configs[2]'s own shape: 1024 queries = FOUR query groups per tile stream (the pacing poll over three partners)
    # and more rows than 16384, so that filter levels of the cascade run; top-100 (M = 225) and top-10
    (768, 40000, 1024, 100, "l2"),
    (768, 40000, 1024, 10, "ip"),
    (768, 40000, 2048, 10, "l2"),           # eight query groups: the most one pacing line serves
])
def test_parity_mfma_regime(coracle, d, n, nq, top_k, metric):
    rows, queries, classes, rng = synth_case(n, d, "bf16", nq, seed=77)
    m = min(nq, 4)
    match = np.zeros((nq, n), np.uint8)
    match[:m] = rng.random((m, n)) < 0.08
    names, texts = names_for_match(match[:m])
    texts = texts + ["plain query"] * (nq - m)
    col = make_collection(rows, "bf16", metric, names, classes)
    col.set_path("mfma")
    check_against_oracle(coracle, col, rows, queries, classes, match, top_k, 2.0, metric, texts)
    st = col.stats()
    assert st["mfma_batches"] >= 1


@pytest.mark.parametrize("d,n,nq,top_k,metric,scan", [
    (1536, 3000, 8, 10, "l2", "gemv"),          # below the switch-over (9 queries): one GEMV group of 8
    (2048, 9000, 2, 10, "ip", "gemv"),
    (2048, 2500, 130, 100, "l2", "mfma"),       # two query tiles of 128, ragged second one; one ragged row tile
    (1100, 3000, 33, 10, "ip", "mfma"),         # padded to 1536
    (1536, 30011, 9, 10, "l2", "mfma"),         # the first batch size that takes it; several cascade levels
    (1536, 20000, 300, 100, "l2", "mfma"),      # three query tiles: workgroups that share a row tile, default loads
    (2048, 127, 40, 10, "l2", "mfma"),          # less than one row tile
    (1536, 129, 64, 200, "ip", "mfma"),         # one row past a tile; top-200 of 129 rows
])
def test_parity_long_bf16_rows(coracle, d, n, nq, top_k, metric, scan):
    """bf16 rows longer than 1024 elements have no query-stationary MFMA scan (the queries do not fit a wave's
    registers): the auto path serves small batches with one GEMV group of up to 8 queries and batches of 9 and more with the
    LDS-tiled scan of crag_scan_qlong.hip (128 rows x 128 queries per workgroup)."""
    rows, queries, classes, rng = synth_case(n, d, "bf16", nq, seed=79)
    match = np.zeros((nq, n), np.uint8)
    match[:2] = rng.random((2, n)) < 0.08
    names, texts = names_for_match(match[:2])
    texts = texts + ["plain query"] * (nq - 2)
    col = make_collection(rows, "bf16", metric, names, classes)
    check_against_oracle(coracle, col, rows, queries, classes, match, top_k, 2.0, metric, texts)
    st = col.stats()
    if scan == "gemv":
        assert st["gemv_batches"] >= 1 and st["mfma_batches"] == 0
    else:
        assert st["mfma_batches"] >= 1 and st["gemv_batches"] == 0
    assert st["fallback"] == 0


@pytest.mark.parametrize("d,nq,scan", [(768, 8, "gemv"), (768, 9, "mfma"), (384, 8, "gemv"), (384, 9, "mfma"),
                                       (384, 33, "mfma"), (384, 64, "mfma"), (384, 65, "mfma"), (384, 96, "mfma"), (384, 97, "mfma"), (128, 20, "mfma"), (128, 4, "gemv"), (190, 5, "mfma"),
                                       (1000, 50, "mfma"), (512, 129, "mfma"), (384, 160, "mfma"), (768, 192, "mfma")])
def test_f32_rows_switch_over(coracle, d, nq, scan):
    """float32 rows: one GEMV group up to 8 queries (4 on rows up to 192 elements), then the LDS-tiled scan (32-, 64-,
    96-, 128-query tiles)."""
    rows, queries, classes, rng = synth_case(7000, d, "f32", nq, seed=83)
    match = np.zeros((nq, 7000), np.uint8)
    match[:2] = rng.random((2, 7000)) < 0.08
    names, texts = names_for_match(match[:2])
    texts = texts + ["plain query"] * (nq - 2)
    col = make_collection(rows, "f32", "l2", names, classes)
    for top_k in (10, 100):
        check_against_oracle(coracle, col, rows, queries, classes, match, top_k, 2.0, "l2", texts)
    st = col.stats()
    assert (st["mfma_batches"] >= 1) == (scan == "mfma") and (st["gemv_batches"] >= 1) == (scan == "gemv")
    assert st["fallback"] == 0


def test_f32_tiled_scan_duplicates_take_the_fallback_tiers(coracle):
    n, d, nq = 6000, 384, 24
    rows, queries, classes, rng = synth_case(n, d, "f32", nq, seed=85)
    rows[100:400] = rows[7]
    queries[:4] = rows[7]
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    st = col.stats()
    assert st["mfma_batches"] >= 1 and st["fallback"] >= 4


def test_long_bf16_rows_duplicates_take_the_fallback_tiers(coracle):
    """Every fallback tier behind the LDS-tiled long-row scan: duplicated vectors at the 2*top_k boundary (tier 1),
    a certified re-scan (tier 1b goes through the same kernel with its refined tau), and a collection of identical rows."""
    n, d, nq = 6000, 1536, 24
    rows, queries, classes, rng = synth_case(n, d, "bf16", nq, seed=81)
    rows[100:400] = rows[7]
    rows[2000:2400] = rows[1999]
    queries[:4] = rows[7]
    queries[4:8] = rows[1999]
    col = make_collection(rows, "bf16", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    st = col.stats()
    assert st["mfma_batches"] >= 1 and st["fallback"] >= 8
    rows[:] = rows[0]
    col2 = make_collection(rows, "bf16", "l2", None, classes)
    check_against_oracle(coracle, col2, rows, queries, classes, None, 10, 2.0, "l2", None)
    assert col2.stats()["fallback"] == nq


@pytest.mark.parametrize("d,n,nq,top_k,metric", [
    (384, 30011, 40, 10, "l2"),             # the reference's MiniLM rows; ragged last tile; two active waves
    (384, 9000, 130, 100, "l2"),            # two query groups, top-100
    (256, 5000, 33, 8, "ip"),
    (100, 3000, 20, 10, "l2"),              # padded to 128
    (190, 4000, 17, 10, "l2"),              # padded to 192
    (384, 40, 9, 10, "l2"),                 # a single ragged tile
    (768, 9000, 40, 10, "l2"),              # longer than 384 elements: the LDS-tiled scan (crag_scan_qlong.hip)
    (512, 5000, 130, 100, "ip"),            # two query tiles
    (1000, 3000, 33, 10, "l2"),             # padded to 1024
    (768, 127, 9, 200, "l2"),               # less than one row tile
])
def test_parity_f32_mfma_regime(coracle, d, n, nq, top_k, metric):
    """float32 rows through the v_mfma_f32_32x32x2_f32 form of the LDS-tiled scan (crag_scan_qlong.hip), forced."""
    rows, queries, classes, rng = synth_case(n, d, "f32", nq, seed=78)
    match = np.zeros((nq, n), np.uint8)
    match[:4] = rng.random((4, n)) < 0.08
    names, texts = names_for_match(match[:4])
    texts = texts + ["plain query"] * (nq - 4)
    col = make_collection(rows, "f32", metric, names, classes)
    col.set_path("mfma")
    check_against_oracle(coracle, col, rows, queries, classes, match, top_k, 2.0, metric, texts)
    assert col.stats()["mfma_batches"] >= 1


@pytest.mark.parametrize("dtype,d,nq,path", [("f32", 384, 40, "mfma"), ("f32", 128, 9, "mfma"), ("f32", 192, 33, "mfma"),
                                              ("f32", 256, 129, "mfma"), ("f32", 384, 3, "gemv"), ("bf16", 768, 8, "gemv"), ("bf16", 768, 40, "mfma"),
                                              ("bf16", 1024, 20, "mfma"), ("f32", 1024, 2, "gemv"), ("bf16", 256, 33, "mfma"),
                                              ("bf16", 768, 150, "mfma"), ("bf16", 384, 129, "mfma"), ("bf16", 512, 256, "mfma"),
                                              ("bf16", 256, 192, "mfma"), ("bf16", 1000, 140, "mfma"), ("bf16", 1024, 9, "mfma"),
                                              ("bf16", 1536, 40, "mfma"), ("bf16", 2048, 130, "mfma"), ("bf16", 1500, 3, "mfma"),
                                              ("f32", 768, 40, "mfma"), ("f32", 1024, 130, "mfma"), ("f32", 500, 3, "mfma"),
                                              # several query groups per tile stream: the sample kernels without nontemporal DMA
                                              ("bf16", 256, 300, "mfma"), ("bf16", 384, 300, "mfma"), ("bf16", 512, 257, "mfma"),
                                              ("bf16", 768, 260, "mfma"),
                                              # the LDS-tiled scan on float32 rows of every length: 32-, 64- and 128-query tiles
                                              ("f32", 384, 9, "tiled"), ("f32", 384, 40, "tiled"), ("f32", 384, 100, "tiled"),
                                              ("f32", 128, 33, "tiled"), ("f32", 190, 20, "tiled"), ("f32", 256, 64, "tiled"),
                                              ("f32", 768, 20, "tiled"), ("f32", 1024, 64, "tiled"), ("f32", 512, 200, "tiled"),
                                              ("f32", 384, 70, "tiled"), ("f32", 768, 96, "tiled"), ("f32", 384, 160, "tiled"),
                                              ("bf16", 1536, 128, "mfma"), ("bf16", 2048, 20, "mfma"), ("bf16", 1536, 90, "mfma"),
                                              # the query-stationary scan at every row length with several query groups, and
                                              # with a single query (three of a wave's four query blocks empty)
                                              ("bf16", 512, 300, "mfma"), ("bf16", 384, 600, "mfma"), ("bf16", 256, 1, "mfma"),
                                              ("bf16", 768, 1100, "mfma")])
def test_scan_error_bound(dtype, d, nq, path):
    """The certificate assumes |approximate scan score - exact score| <= eps with
    eps = 2 * 2^-24 * ((dim_pad + 4) (2 |q||x|max + |x|max^2) + 3 |x|max^2).  Check the scans against fp64 on every
    (query, row) pair -- this also pins every lane->(row, query) mapping of both scan kernels."""
    import torch
    n = 2311
    rows = O.synth_unit_rows(n, d, 1, dtype)
    qs = O.synth_unit_rows(nq, d, 2, dtype)
    col = make_collection(rows, dtype)
    q = torch.from_numpy(qs).cuda() if dtype == "f32" else torch.from_numpy(qs.view(np.int16)).cuda().view(torch.bfloat16)
    got = col.debug_dense_scores(q, path).cpu().numpy().astype(np.float64)
    X, Q = O.as_f64(rows), O.as_f64(qs)
    ref = (X * X).sum(1)[None, :] - 2.0 * Q @ X.T
    xmax = float((X * X).sum(1).max())
    qn = np.sqrt((Q * Q).sum(1))
    eps = 2.0 * 2.0 ** -24 * ((col.dim_pad + 4) * (2.0 * qn * np.sqrt(xmax) + xmax) + 3.0 * xmax)
    err = np.abs(got - ref).max(axis=1)
    assert np.all(err <= eps), (err.max(), eps.min())
    assert err.max() < 2e-6          # typical error is ~1e-7: the bound has three orders of margin


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_gemv_every_compiled_variant(dtype):
    """Every k_scan_gemv<DT, C, NQ> the library compiles (C = 16-byte chunks per row: 32..256, NQ = queries per
    lane group: 1, 2, 4, 8), each scored on every (query, row) pair against fp64.  The variants are separate
    instruction schedules: round 2's fuzz found <bf16, 192, 1> and <bf16, 256, 1> reading an accumulator too soon
    after its last v_dot2c (wrong score for the last row of a 64-row tile) while every other variant was right."""
    import torch
    per16 = 8 if dtype == "bf16" else 4
    worst = 0.0
    for chunks in (32, 48, 64, 96, 128, 192, 256):
        d = chunks * per16
        if dtype == "f32" and d > 1024:
            continue
        for d_use in (d, d - 3 * per16 - 1):       # the padded tail of a row too
            rows = O.synth_unit_rows(1500, d_use, chunks, dtype)
            qs = O.synth_unit_rows(8, d_use, chunks + 1, dtype)
            col = make_collection(rows, dtype)
            assert col.dim_pad * (2 if dtype == "bf16" else 4) // 16 == chunks
            X, Q = O.as_f64(rows), O.as_f64(qs)
            ref = (X * X).sum(1)[None, :] - 2.0 * Q @ X.T
            for nq in (1, 2, 3, 8):
                q = torch.from_numpy(qs[:nq]).cuda() if dtype == "f32" else \
                    torch.from_numpy(qs[:nq].view(np.int16)).cuda().view(torch.bfloat16)
                for rep in range(2):
                    got = col.debug_dense_scores(q, "gemv").cpu().numpy().astype(np.float64)
                    err = np.abs(got - ref[:nq])
                    bad = np.argwhere(err > 2e-6)
                    assert len(bad) == 0, (dtype, chunks, d_use, nq, float(err.max()), bad[:6].tolist())
                    worst = max(worst, float(err.max()))
            col.close()
    assert worst < 2e-6


@pytest.mark.parametrize("d,n,nq", [(1536, 30011, 1), (2048, 9000, 1), (1536, 9000, 2), (1100, 9000, 1)])
def test_parity_long_bf16_rows_few_queries(coracle, d, n, nq):
    """Rows longer than 1024 elements with one or two queries: the 192- and 256-chunk GEMV variants end to end."""
    rows, queries, classes, _ = synth_case(n, d, "bf16", nq, seed=d + nq)
    col = make_collection(rows, "bf16", "l2", None, classes)
    for top_k in (10, 200):
        check_against_oracle(coracle, col, rows, queries, classes, None, top_k, 2.0, "l2", None)
    assert col.stats()["fallback"] == 0


@pytest.mark.parametrize("n", [1, 2, 5, 19, 20, 21, 63, 64, 65, 127, 129, 4095, 4097])
def test_small_and_ragged_collections(coracle, n):
    rows, queries, classes, _ = synth_case(n, 384, "f32", 2, seed=5)
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)


@pytest.mark.parametrize("dtype,d", [("bf16", 768), ("f32", 384)])
def test_batches_larger_than_a_chunk(coracle, dtype, d):
    """More queries than the library launches at once (2048): consecutive chunks on one stream."""
    rows, queries, classes, rng = synth_case(3000, d, dtype, 2100, seed=55)
    col = make_collection(rows, dtype, "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 5, 2.0, "l2", None)


def test_empty_collection_returns_nothing():
    col = ch.GpuCollection(64, "f32")
    col.finalize()
    ids, sc, di, cn = col.search(np.ones((2, 64), np.float32), 5)
    assert list(cn) == [0, 0] and np.all(ids == -1)


def test_symbol_boost_threshold_and_penalties(coracle):
    """R4: with symbol_boost < 0.5 the > 0.01 filter bites; penalised rows are overtaken."""
    rows, queries, classes, rng = synth_case(6000, 384, "f32", 3, seed=21)
    match = (rng.random((3, 6000)) < 0.5).astype(np.uint8)
    names, texts = names_for_match(match)
    col = make_collection(rows, "f32", "l2", names, classes)
    for sb in (0.004, 0.02, 1.0, 2.5):
        check_against_oracle(coracle, col, rows, queries, classes, match, 10, sb, "l2", texts)


def test_exact_ties_and_duplicates_use_row_id_order(coracle):
    base = O.synth_unit_rows(7, 384, 3, "f32")
    rows = np.tile(base, (300, 1))                       # every vector 300 times: massive exact ties
    queries = O.synth_unit_rows(2, 384, 4, "f32")
    classes = np.zeros(len(rows), np.uint8)
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)


@pytest.mark.parametrize("nq", [131, 9, 3])
def test_mfma_massive_ties_overflow_segments_and_fall_back(coracle, nq):
    """bf16 through the segmented, atomic-free candidate buffers of the two-block MFMA scan (two query groups,
    a few queries, three queries): every vector 3000 times, so thousands of rows tie at each query's
    threshold, the per-workgroup segments overflow, and those queries must come back exact through the fallback."""
    base = O.synth_unit_rows(7, 768, 3, "bf16")
    rows = np.tile(base, (3000, 1))
    queries = O.synth_unit_rows(max(nq, 7), 768, 4, "bf16")
    queries[:7] = base                                    # distance-0 ties as well
    queries = np.ascontiguousarray(queries[:nq])
    classes = np.zeros(len(rows), np.uint8)
    col = make_collection(rows, "bf16", "l2", None, classes)
    col.set_path("mfma")
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    st = col.stats()
    assert st["mfma_batches"] >= 1 and st["fallback"] >= 1


@pytest.mark.parametrize("dtype,d,nq,path", [("bf16", 768, 40, "mfma"), ("f32", 384, 3, "gemv"), ("bf16", 1024, 20, "mfma"), ("f32", 384, 40, "mfma")])
def test_duplicated_chunks_are_settled_by_rescoring_the_buffer(coracle, dtype, d, nq, path):
    """Vendored copies and licence headers: a chunk that sits at a query's 2*top_k boundary exists 60 times, so the
    slack of 16 re-scored rows cannot separate it and the certificate fails -- but the candidate buffer holds every
    copy, so re-scoring the whole buffer exactly (fallback tier 1) settles the query without the full exact scan."""
    n, top_k = 40000, 10
    rows, queries, classes, rng = synth_case(n, d, dtype, nq, seed=83)
    near = coracle.knn(rows, queries, 2 * top_k, "l2")[0]
    for b in range(min(nq, 3)):                                   # the 15th nearest of the first queries, 60 copies each
        src = int(near[b, 14])
        where = rng.choice(np.arange(20000, n), size=60, replace=False)
        rows[where] = rows[src]
    col = make_collection(rows, dtype, "l2", None, classes)
    col.set_path(path)
    check_against_oracle(coracle, col, rows, queries, classes, None, top_k, 2.0, "l2", None)
    st = col.stats()
    assert st["fallback"] >= min(nq, 3) and st["exact_scan"] == 0, st


@pytest.mark.parametrize("dtype,d,nq,path", [("f32", 384, 33, "auto"), ("bf16", 768, 140, "mfma"), ("f32", 384, 5, "gemv"),
                                              ("bf16", 1024, 9, "mfma")])
def test_whole_batch_of_massive_ties_is_settled_by_the_certified_rescan(coracle, dtype, d, nq, path):
    """Every vector exists 3000 times and the queries are corpus vectors: each query's 2*top_k nearest are decided by
    row id among 3000 exact ties and no certificate of the first pass can hold.  One more scan of the whole batch with
    thresholds derived from the distances already known (fallback tier 1b) collects every tied row, and re-scoring
    them exactly settles all queries -- no full exact scan."""
    base = O.synth_unit_rows(160, d, 5, dtype)
    rows = np.tile(base, (3000, 1))
    queries = np.ascontiguousarray(base[:nq])
    classes = (np.arange(len(rows)) % 3).astype(np.uint8)
    col = make_collection(rows, dtype, "l2", None, classes)
    col.set_path(path)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    st = col.stats()
    assert st["fallback"] == nq and st["rescan"] == nq and st["exact_scan"] == 0, st


def test_whole_batch_through_the_exact_scan(coracle):
    """More exact ties than a candidate buffer holds (three distinct vectors, 20000 copies each): every query of the
    batch needs the full exact scan at once, and the batched device-side radix select must answer all of them."""
    base = O.synth_unit_rows(3, 384, 5, "f32")
    rows = np.tile(base, (20000, 1))
    queries = np.concatenate([base, O.synth_unit_rows(30, 384, 6, "f32")])
    classes = (np.arange(len(rows)) % 3).astype(np.uint8)
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    st = col.stats()
    assert st["exact_scan"] == 33 and st["fallback"] == 33, st


def test_fallback_inside_a_multi_collection_batch(coracle):
    """search_multi where SOME collections force the exact scan (all-duplicate rows) and others do not: the failed
    queries' own collections (different sizes) go through the batched fallback, the rest are untouched."""
    d, top_k = 128, 5
    cols, data = [], []
    for i, (n, dup) in enumerate([(3000, False), (20000, True), (900, True), (12000, False), (7, True)]):
        rows, _, classes, _ = synth_case(n, d, "f32", 1, seed=400 + i)
        if dup:
            rows[:] = rows[0]                                    # every row identical: ties decide everything
        cols.append(make_collection(rows, "f32", "l2", None, classes, id_base=100_000 * i))
        data.append((rows, classes))
    order = [0, 1, 1, 2, 3, 4, 3]
    q = O.synth_unit_rows(len(order), d, 998, "f32")
    ids, sc, di, cn = ch.search_multi([cols[i] for i in order], q, top_k, 2.0)
    for b, i in enumerate(order):
        rows, classes = data[i]
        (wi, ws, wd), = coracle.search_numeric(rows, q[b:b + 1], classes, None, top_k, 2.0, "l2", 100_000 * i)
        k = len(wi)
        assert int(cn[b]) == k and np.array_equal(ids[b, :k], wi) and np.array_equal(sc[b, :k], ws), (b, i)
    st = cols[0].stats()                                          # counted on the first handle of the call
    # 20000 identical rows overflow the candidate buffer (full exact scan, two queries); 900 identical rows fit their
    # buffer whole, so re-scoring the buffer settles that query; 7 rows are all re-scored by the first pass already
    assert st["fallback"] == 3 and st["exact_scan"] == 2, st


def test_near_ties_force_the_exact_fallback(coracle):
    """Rows that differ from each other by one float32 ulp in one coordinate: the approximate scan
    cannot separate them, the certificate must fail and the exact fallback must still agree."""
    rng = np.random.default_rng(9)
    n, d = 5000, 384
    q = O.synth_unit_rows(1, d, 10, "f32")
    rows = np.tile(q, (n, 1)).astype(np.float32)
    col_idx = rng.integers(0, d, n)
    for i in range(n):
        v = rows[i, col_idx[i]]
        rows[i, col_idx[i]] = np.nextafter(v, np.float32(np.inf) if i % 2 else np.float32(-np.inf))
    rows[::7] *= np.float32(1.0000001)
    classes = np.zeros(n, np.uint8)
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, q, classes, None, 10, 2.0, "l2", None)
    assert col.stats()["fallback"] >= 1


def test_stage1_entries_equal_oracle_candidates(coracle):
    import torch
    rows, queries, classes, rng = synth_case(12000, 768, "bf16", 4, seed=31)
    match = (rng.random((4, 12000)) < 0.1).astype(np.uint8)
    names, texts = names_for_match(match)
    col = make_collection(rows, "bf16", "l2", names, classes, id_base=1000)
    top_k, kc = 10, 20
    q = torch.from_numpy(queries.view(np.int16)).cuda().view(torch.bfloat16)
    hashes = [col.symbol_hashes(t) for t in texts]
    off = np.zeros(5, np.int32)
    np.cumsum([len(h) for h in hashes], out=off[1:])
    qh = torch.from_numpy(np.concatenate(hashes).view(np.int64)).cuda()
    qo = torch.from_numpy(off).cuda()
    ent = torch.zeros((4, kc, 16), dtype=torch.uint8, device="cuda")
    col.search_local_device(q, top_k, ent, qh, qo)
    e = ent.cpu().numpy().reshape(4, kc * 16).view(ENTRY)
    ids, dist, _ = coracle.knn(rows, queries, kc, "l2", id_base=1000)
    assert np.array_equal(e["id"], ids)
    assert np.array_equal(e["dist"].view(np.uint32), dist.view(np.uint32))
    for b in range(4):
        pos = ids[b] - 1000
        assert np.array_equal(e["flags"][b] & 3, classes[pos])
        assert np.array_equal((e["flags"][b] >> 2) & 1, match[b][pos])


@pytest.mark.parametrize("n_shards", [2, 3, 8])
def test_sharded_merge_equals_single_collection(coracle, n_shards):
    """Multi-GPU path on one GPU: G shard collections with global id bases, stage 1 on each, the
    gathered [G, B, kc, 16] entries merged by the stage-2 kernel."""
    import torch
    from coderag_hip.dist import shard_range
    n, d, nq, top_k = 20001, 768, 5, 10
    rows, queries, classes, rng = synth_case(n, d, "bf16", nq, seed=41)
    match = (rng.random((nq, n)) < 0.1).astype(np.uint8)
    names, texts = names_for_match(match)
    q = torch.from_numpy(queries.view(np.int16)).cuda().view(torch.bfloat16)
    ents = torch.zeros((n_shards, nq, 2 * top_k, 16), dtype=torch.uint8, device="cuda")
    cols = []
    for r in range(n_shards):
        lo, hi = shard_range(n, r, n_shards)
        c = make_collection(rows[lo:hi], "bf16", "l2", names[lo:hi], classes[lo:hi], id_base=lo)
        cols.append(c)
        hashes = [c.symbol_hashes(t) for t in texts]
        off = np.zeros(nq + 1, np.int32)
        np.cumsum([len(h) for h in hashes], out=off[1:])
        cat = np.concatenate(hashes) if sum(map(len, hashes)) else np.zeros(1, np.uint64)
        c.search_local_device(q, top_k, ents[r], torch.from_numpy(cat.view(np.int64)).cuda(),
                              torch.from_numpy(off).cuda())
    ids, sc, di, cn = ch.merge_rerank_device(0, ents, n_shards, nq, top_k, 2.0)
    torch.cuda.synchronize()
    want = coracle.search_numeric(rows, queries, classes, match, top_k, 2.0, "l2")
    for b, (wi, ws, wd) in enumerate(want):
        k = len(wi)
        assert int(cn[b]) == k
        assert np.array_equal(ids[b, :k].cpu().numpy(), wi)
        assert np.array_equal(sc[b, :k].cpu().numpy(), ws)
        assert np.array_equal(di[b, :k].cpu().numpy().view(np.uint32), wd.view(np.uint32))


def test_workspace_isolated_collections_queried_together(coracle):
    """C5 in miniature: every query addresses its own collection; a query only sees its own rows."""
    d, top_k = 1024, 8
    sizes = [3000, 1, 700, 4500, 64, 2500, 2500]
    cols, data, qs = [], [], []
    for i, n in enumerate(sizes):
        rows, queries, classes, _ = synth_case(n, d, "bf16", 1, seed=100 + i)
        cols.append(make_collection(rows, "bf16", "l2", None, classes))
        data.append((rows, classes))
        qs.append(queries[0])
    order = [0, 0, 1, 2, 3, 3, 3, 4, 5, 6, 0]
    q = np.stack([qs[i] for i in order])
    ids, sc, di, cn = ch.search_multi([cols[i] for i in order], to_f32(q), top_k, 2.0)
    for b, i in enumerate(order):
        rows, classes = data[i]
        (wi, ws, wd), = coracle.search_numeric(rows, q[b:b + 1], classes, None, top_k, 2.0, "l2")
        k = len(wi)
        assert int(cn[b]) == k and np.array_equal(ids[b, :k], wi) and np.array_equal(sc[b, :k], ws)


def test_multi_collection_device_entry_point_equals_host_one():
    """crag_search_multi_device (queries and symbol hashes resident in HBM) against crag_search_multi."""
    import torch
    rng = np.random.default_rng(12)
    cols, texts = [], []
    for c in range(5):
        rows, _, classes, _ = synth_case(3000 + 777 * c, 1024, "bf16", 1, seed=100 + c)
        match = (rng.random((1, len(rows))) < 0.1).astype(np.uint8)
        names, tx = names_for_match(match)
        cols.append(make_collection(rows, "bf16", "l2", names, classes, id_base=10_000 * c))
        texts.append(tx[0])
    order = [0, 3, 1, 4, 2, 3]                       # a handle may repeat
    qbits = O.synth_unit_rows(len(order), 1024, 77, "bf16")
    sel = [cols[i] for i in order]
    tx = [texts[i] for i in order]
    want = ch.search_multi(sel, O.bf16_bits_to_f32(qbits), 8, 2.0, tx)
    hashes = [c.symbol_hashes(t) for c, t in zip(sel, tx)]
    off = np.zeros(len(order) + 1, np.int32)
    np.cumsum([len(h) for h in hashes], out=off[1:])
    qh = torch.from_numpy(np.concatenate(hashes).view(np.int64)).cuda()
    qo = torch.from_numpy(off).cuda()
    q = torch.from_numpy(qbits.view(np.int16)).cuda().view(torch.bfloat16)
    got = ch.search_multi_device(sel, q, 8, 2.0, qh, qo)
    for a, b in zip(got, want):
        assert np.array_equal(a.cpu().numpy(), b)


@pytest.mark.parametrize("nq", [2, 64])
def test_multi_collection_mixed_sizes_with_and_without_level1(coracle, nq):
    """One grouped call over collections of very different sizes: some are smaller than the level-0 sample, some
    have no level 1, some have one (a level-1 selection launched for the whole batch must leave the seeds of the
    queries that had no level 1 alone).  Both multi-collection entry points against the oracle."""
    import torch
    d, top_k = 128, 10
    sizes = [7, 3000, 8192, 10_000, 40_000, 320_000]
    cols, data = [], []
    for i, n in enumerate(sizes):
        rows, _, classes, _ = synth_case(n, d, "f32", 1, seed=300 + i)
        cols.append(make_collection(rows, "f32", "l2", None, classes, id_base=1_000_000 * i))
        data.append((rows, classes))
    order = [5, 1] if nq == 2 else [(3 * b) % len(sizes) for b in range(nq - 4)] + [5, 1, 0, 2]
    order = sorted(order)  # consecutive queries on one collection share a group
    q = O.synth_unit_rows(len(order), d, 999, "f32")
    sel = [cols[i] for i in order]
    got_host = ch.search_multi(sel, q, top_k, 2.0)
    got_dev = [t.cpu().numpy() for t in ch.search_multi_device(sel, torch.from_numpy(q).cuda(), top_k, 2.0)]
    for b, i in enumerate(order):
        rows, classes = data[i]
        (wi, ws, wd), = coracle.search_numeric(rows, q[b:b + 1], classes, None, top_k, 2.0, "l2", 1_000_000 * i)
        k = len(wi)
        for ids, sc, di, cn in (got_host, got_dev):
            assert int(cn[b]) == k, (b, i, int(cn[b]), k)
            assert np.array_equal(ids[b, :k], wi), (b, i)
            assert np.array_equal(sc[b, :k], ws) and np.array_equal(di[b, :k].view(np.uint32), wd.view(np.uint32)), (b, i)
    assert sum(c.stats()["fallback"] for c in cols) == 0


def test_concurrent_searches_on_one_handle(coracle):
    rows, queries, classes, _ = synth_case(20000, 384, "f32", 8, seed=51)
    col = make_collection(rows, "f32", "l2", None, classes)
    want = coracle.search_numeric(rows, queries, classes, None, 10, 2.0, "l2")
    errs = []

    def work(b):
        try:
            for _ in range(5):
                ids, sc, di, cn = col.search(queries[b:b + 1], 10, 2.0)
                assert np.array_equal(ids[0, :int(cn[0])], want[b][0]) and np.array_equal(sc[0, :int(cn[0])], want[b][1])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=work, args=(b,)) for b in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_c2_full_size_batch1(coracle):
    """configs[1]: 1M x 384 fp32, batch-1, top-10 -- full size, against the C oracle."""
    rows = O.synth_unit_rows(1_000_000, 384, 1234, "f32")
    queries = O.synth_unit_rows(2, 384, 1235, "f32")
    classes = np.zeros(len(rows), np.uint8)
    col = make_collection(rows, "f32", "l2", None, classes)
    check_against_oracle(coracle, col, rows, queries, classes, None, 10, 2.0, "l2", None)
    assert col.stats()["fallback"] == 0



class _LazyTable:
    """Indexable stand-in for a 156 250-entry Python list (ids / metadatas / documents of a ChunkCollection)."""

    def __init__(self, n, fn):
        self._n, self._fn = n, fn

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._fn(int(i))


def test_c5_full_size_64_collections_against_the_oracle(coracle, tmp_path):
    """configs[4] at its full size -- 64 workspace-isolated collections x 156 250 rows x 1024 bf16 (10M rows in all) with
    Zipf-drawn chunk names and path classes as bench.py builds them, one query per collection, each naming two symbols
    (one of them the name of a row among the query's own nearest, so that the boost reorders the result) -- through
    crag_search_multi_device and through HybridSearcher.search_many, against the C oracle per collection."""
    import torch
    from coderag_hip.hybrid_search import ChunkCollection  # noqa: F401  (the class search_many serves)
    ncol, nper, d, top_k, n_sym = 64, 156_250, 1024, 8, 50_000
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(4321)
    sym_names = [f"sym{i}" for i in range(n_sym)]
    sym_hash, sym_len = ch.hash_names(sym_names)
    sym_hash_t = torch.from_numpy(sym_hash.view(np.int64)).to(dev)
    bucket = {int(h): {nm} for h, nm in zip(sym_hash, sym_names)}
    rng = np.random.default_rng(5)
    x = torch.randn((ncol, d), generator=g, device=dev, dtype=torch.float32)
    q_dev = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
    q_bits = q_dev.view(torch.int16).cpu().numpy().view(np.uint16)
    cols, texts, want, host = [], [], [], []
    for c in range(ncol):
        x = torch.randn((nper, d), generator=g, device=dev, dtype=torch.float32)
        rr = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16).contiguous()
        zipf = (torch.rand((nper,), generator=g, device=dev).pow(3.0) * n_sym).long().clamp_(max=n_sym - 1)
        nh = sym_hash_t[zipf].contiguous()
        u = torch.rand((nper,), generator=g, device=dev)
        pc = torch.where(u < 0.70, 0, torch.where(u < 0.95, 2, 1)).to(torch.uint8).contiguous()
        col = ch.GpuCollection(d, "bf16", "l2")
        col.attach_device(rr, nh, pc)
        col.set_symbols(sym_hash, sym_len)
        col.finalize()
        cols.append(col)
        rows_h = rr.view(torch.int16).cpu().numpy().view(np.uint16)
        zipf_h, pc_h = zipf.cpu().numpy(), pc.cpu().numpy()
        near = coracle.knn(rows_h, q_bits[c:c + 1], 2 * top_k, "l2")[0][0]
        pick = next(int(r) for r in near[3:] if pc_h[r] != 1)     # a candidate behind the third that no path penalty hides
        text = f"Where is {sym_names[int(zipf_h[pick])].upper()} used together with sym{int(rng.integers(0, n_sym))}?"
        texts.append(text)
        named = np.array([i for i, nm in enumerate(sym_names) if nm in text.lower()], np.int64)   # `name in q_lower`
        assert len(named) >= 2
        match = np.isin(zipf_h, named)[None, :].astype(np.uint8)
        (wi, ws, wd), = coracle.search_numeric(rows_h, q_bits[c:c + 1], pc_h, match, top_k, 2.0, "l2")
        assert match[0][wi].any()                                  # a boosted row made it into the answer
        want.append((wi, ws, wd))
        host.append((zipf_h, pc_h))
        del rows_h, x
    # (1) the device-buffer entry point: queries and their symbol hashes resident in HBM
    hashes = [cols[c].symbol_hashes(texts[c]) for c in range(ncol)]
    off = np.zeros(ncol + 1, np.int32)
    np.cumsum([len(h) for h in hashes], out=off[1:])
    qh = torch.from_numpy(np.concatenate(hashes).view(np.int64)).to(dev)
    qo = torch.from_numpy(off).to(dev)
    ids, sc, di, cn = [t.cpu().numpy() for t in ch.search_multi_device(cols, q_dev, top_k, 2.0, qh, qo)]
    for c, (wi, ws, wd) in enumerate(want):
        k = len(wi)
        assert int(cn[c]) == k and np.array_equal(ids[c, :k], wi), c
        assert np.array_equal(sc[c, :k], ws) and np.array_equal(di[c, :k].view(np.uint32), wd.view(np.uint32)), c
    # (2) the reference's class: one search_many call = one grouped launch over the 64 workspaces
    qf = O.bf16_bits_to_f32(q_bits)
    by_text = {t: qf[c] for c, t in enumerate(texts)}
    hs = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=lambda ts: np.stack([by_text[t] for t in ts]), dtype="bf16")
    names = []
    for c in range(ncol):
        cc = hs.create_collection(f"code_chunks__ws{c:02d}", d, dtype="bf16")
        cc.gpu.close()
        cc.gpu = cols[c]
        zipf_h, pc_h = host[c]
        cc.ids = _LazyTable(nper, lambda i, c=c: f"{c:02d}-{i}")
        cc.metadatas = _LazyTable(nper, lambda i, z=zipf_h: {"name": sym_names[int(z[i])]})
        cc.documents = _LazyTable(nper, lambda i: None)
        cc._bucket = bucket
        names.append(cc.name)
    got = hs.search_many(texts, names, top_k=top_k, symbol_boost=2.0)
    for c, (wi, ws, wd) in enumerate(want):
        assert [r["id"] for r in got[c]] == [f"{c:02d}-{int(i)}" for i in wi], c
        assert [r["score"] for r in got[c]] == [float(s) for s in ws], c
        assert all(r["metadata"]["name"] == sym_names[int(host[c][0][int(i)])] for r, i in zip(got[c], wi))
    assert sum(col.stats()["fallback"] for col in cols) == 0
    for col in cols:
        col.close()


def test_full_size_10m_bf16_properties():
    """BASELINE.json's metric corpus (10M x 768 bf16) is too big for the CPU oracle in a test, so
    check size-independent properties: planted exact copies of the query come back first with
    distance 0 in ascending id order, distances ascend, ids are distinct, and the reported
    distances agree with an fp64 recomputation of the returned rows."""
    import torch
    n, d, nq = 10_000_000, 768, 1024
    g = torch.Generator(device="cuda").manual_seed(1234)
    rows = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    for lo in range(0, n, 1_000_000):
        x = torch.randn((1_000_000, d), generator=g, device="cuda", dtype=torch.float32)
        rows[lo:lo + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    q = torch.randn((nq, d), generator=g, device="cuda", dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    planted = [9_999_999, 5, 7_654_321]
    for p in planted:
        rows[p] = q[0]
    col = ch.GpuCollection(d, "bf16", "l2")
    col.attach_device(rows)
    col.finalize()
    first4 = {}
    # two-block MFMA cascade with two queries, with one active wave (B = 32) and with a ragged second query group
    # (B = 288), GEMV (B = 4)
    # (B = 288), configs[2] itself (B = 1024 at top-100: four paced query groups, six filter levels), GEMV (B = 4)
    for path, nb, top_k in (("mfma", 2, 10), ("mfma", 32, 10), ("mfma", 288, 10), ("mfma", 1024, 100), ("mfma", 1024, 10),
                            ("gemv", 4, 10)):
        col.set_path(path)
        qq = q[:nb].contiguous()
        ids, sc, di, cn = [t.cpu().numpy() for t in col.search_device(qq, top_k, 2.0)]
        # no metadata columns here, so the score falls with the distance and a top-100 answer starts with the top-10 one
        first4[(path, nb, top_k)] = (ids[:4, :10].copy(), di[:4, :10].copy(), sc[:4, :10].copy())
        assert list(ids[0, :3]) == sorted(planted) and np.all(di[0, :3] == 0.0) and np.all(sc[0, :3] == 1.0)
        for b in range(len(qq)):
            assert int(cn[b]) == top_k and len(set(ids[b])) == top_k
            assert np.all(np.diff(di[b]) >= 0)
            r = rows[torch.from_numpy(ids[b]).cuda()].double()
            ref = ((qq[b].double()[None, :] - r) ** 2).sum(1).cpu().numpy()
            assert np.allclose(di[b], ref, rtol=1e-6, atol=1e-7)
            assert np.allclose(sc[b], 1.0 / (1.0 + di[b].astype(np.float64)), rtol=0, atol=0)
        # the 10th distance must not exceed the 10th smallest of an independent fp32 brute force
        approx = torch.cdist(qq[:2].float(), rows[:2_000_000].float()).pow(2).topk(top_k, largest=False).values
        assert np.all(di[:2, -1] <= approx[:, -1].cpu().numpy() + 1e-5)
    # three independent scan kernels must agree bit for bit on the queries they share
    ref = first4[("gemv", 4, 10)]
    for key, got in first4.items():
        for a, b in zip(got, ref):
            n_common = min(len(a), len(b))
            assert np.array_equal(a[:n_common], b[:n_common]), key
    assert col.stats()["fallback"] == 0


def test_full_size_100m_bf16_one_gpu_properties():
    """configs[3] at one GPU: the whole 100M x 768 bf16 corpus (153.6 GB) on a single MI355X.  Byte offsets
    pass 2^32 and 2^37, so planted copies of a query sit right behind those boundaries and at both ends; the
    three scan kernels must find them (distance 0, ascending id) and agree bit for bit."""
    import torch
    n, d, top_k = 100_000_000, 768, 10
    free, _ = torch.cuda.mem_get_info()
    if free < 175 * 2**30:
        pytest.skip("needs 175 GiB of free HBM")
    g = torch.Generator(device="cuda").manual_seed(4321)
    rows = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    for lo in range(0, n, 1_000_000):
        x = torch.randn((1_000_000, d), generator=g, device="cuda", dtype=torch.float32)
        rows[lo:lo + 1_000_000] = (x / x.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    del x
    q = torch.randn((160, d), generator=g, device="cuda", dtype=torch.float32)
    q = (q / q.norm(dim=1, keepdim=True)).to(torch.bfloat16)
    planted = sorted([0, 2_796_203, 89_478_486, 50_000_000, n - 1])  # 2_796_203 * 1536 > 2^32, 89_478_486 * 1536 > 2^37
    for p in planted:
        rows[p] = q[0]
    col = ch.GpuCollection(d, "bf16", "l2")
    col.attach_device(rows)
    col.finalize()
    assert len(col) == n
    got = {}
    for path, nb in (("mfma", 16), ("mfma", 160), ("gemv", 2)):
        col.set_path(path)
        qq = q[:nb].contiguous()
        ids, sc, di, cn = [t.cpu().numpy() for t in col.search_device(qq, top_k, 2.0)]
        got[(path, nb)] = (ids[:2].copy(), di[:2].copy(), sc[:2].copy())
        assert list(ids[0, :5]) == planted and np.all(di[0, :5] == 0.0) and np.all(sc[0, :5] == 1.0)
        for b in range(min(nb, 8)):
            assert int(cn[b]) == top_k and len(set(ids[b])) == top_k
            assert np.all(np.diff(di[b]) >= 0)
            r = rows[torch.from_numpy(ids[b]).cuda()].double()
            ref = ((qq[b].double()[None, :] - r) ** 2).sum(1).cpu().numpy()
            assert np.allclose(di[b], ref, rtol=1e-6, atol=1e-7)
        # the 10th distance must not exceed the 10th smallest over an independently scored slice at the far end
        tail = rows[n - 2_000_000:].float()
        approx = torch.cdist(qq[1:2].float(), tail).pow(2).topk(top_k, largest=False).values
        assert di[1, -1] <= float(approx[0, -1]) + 1e-5
        del tail
    ref = got[("gemv", 2)]
    for key, val in got.items():
        for a, b in zip(val, ref):
            assert np.array_equal(a, b), key
    assert col.stats()["fallback"] == 0
    col.close()


# ---------------------------------------------------------------------------------------------
def _chunks(n, rng):
    words = ["chunk", "asciiToArray", "stringSize", "hasUnicode", "Parser", "baseFlatten", "debounce", "map", "a",
             "größe", "isArrayLike", "castPath", ""]
    dirs = ["lodash/src/", "src/", "pkg/test/", "lib/", "a/lib/", "types/", "x/__tests__/", "docs/spec/", ""]
    metas, ids, docs = [], [], []
    for i in range(n):
        w = words[int(rng.integers(0, len(words)))]
        p = dirs[int(rng.integers(0, len(dirs)))] + (f"f{i % 17}.d.ts" if i % 11 == 0 else f"f{i % 17}.js")
        m = {"name": w, "path": p, "kind": "Function" if i % 2 else "Class", "start_line": i, "end_line": i + 3}
        metas.append(None if i % 97 == 0 else m)
        ids.append(f"{i:012x}")
        docs.append(f"function {w}() {{\n  return {i};\n}}")
    return ids, metas, docs


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_hybrid_searcher_end_to_end(tmp_path, dtype):
    rng = np.random.default_rng(61)
    n = 3000
    ids, metas, docs = _chunks(n, rng)
    emb = ch.HashingEmbedder()
    vecs = emb(docs)
    hs = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb, dtype=dtype)
    col = hs.create_collection("code_chunks__ws1", 384)
    for lo in range(0, n, 100):                                    # batches of 100, embed_ingest.py:90-105
        col.add(ids[lo:lo + 100], vecs[lo:lo + 100], metas[lo:lo + 100], docs[lo:lo + 100])
    col.gpu.finalize()
    rows = vecs if dtype == "f32" else O.f32_to_bf16_bits(vecs)
    queries = ["Find the implementation of lodash chunk", "What does asciiToArray do?",
               "How to count unicode symbols length in lodash? stringSize hasUnicode", "GRÖSSE größe Parser", "zzz"]
    for q in queries:
        qv = emb([q])[0]
        qv = qv if dtype == "f32" else O.f32_to_bf16_bits(qv)
        for top_k, sb, inc in ((8, 2.0, False), (6, 2.5, True), (3, 0.3, False)):
            got = hs.search(q, top_k=top_k, symbol_boost=sb, include_documents=inc, collection_name="code_chunks__ws1")
            want = O.oracle_search(qv, q, rows, ids, metas, docs, top_k, sb, inc)
            assert got == want
    assert hs.search("x", collection_name="code_chunks__nope") == []
    # /search response shape + post filters (api/main.py:151-216)
    r = ch.search_route(hs, queries[0], top_k=8, x_workspace_id="ws1", path_prefix="lodash/")
    w = O.search_response(queries[0], O.oracle_search(emb([queries[0]])[0] if dtype == "f32" else
                                                      O.f32_to_bf16_bits(emb([queries[0]])[0]),
                                                      queries[0], rows, ids, metas, docs, 8, 2.0, False),
                          path_prefix="lodash/")
    assert r == w
    # search_by_symbol: exact branch and fallback branch (hybrid_search.py:128-146)
    got = hs.search_by_symbol("Parser", top_k=4, collection_name="code_chunks__ws1")
    assert got == O.search_by_symbol("Parser", ids, metas, docs, 4)
    fb = hs.search_by_symbol("nonexistentSymbol", top_k=4, collection_name="code_chunks__ws1")
    assert fb == hs.search("nonexistentSymbol", top_k=4, include_documents=True, collection_name="code_chunks__ws1")
    # persistence round trip (store.py stands where data/chroma_db stands)
    from coderag_hip import store
    store.save_collection(str(tmp_path), col)
    hs2 = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb, dtype=dtype)
    assert hs2.search(queries[0], 8, collection_name="code_chunks__ws1") == hs.search(queries[0], 8, collection_name="code_chunks__ws1")
    batch = hs.search_batch(queries, top_k=8, collection_name="code_chunks__ws1")
    assert batch == [hs.search(q, 8, collection_name="code_chunks__ws1") for q in queries]
    # the same over HTTP (api/main.py:166-232 -> coderag_hip/http_app.py)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from fastapi.testclient import TestClient
        client = TestClient(ch.create_app(hs))
        resp = client.post("/search", json={"query": queries[0], "top_k": 8, "path_prefix": "lodash/"}, headers={"X-Workspace-Id": "ws1"})
        assert resp.status_code == 200 and resp.json() == r
        resp = client.get("/search/Parser", params={"top_k": 4}, headers={"X-Workspace-Id": "ws1"})
        assert resp.json() == {"symbol": "Parser", "total": len(got), "results": got}


def test_search_many_places_collections_and_groups_launches(tmp_path):
    """64-workspace serving in miniature: a searcher with two placement slots (both GPU 0 here: a one-GPU box stands
    in for two devices), collections placed round-robin, one search_many call = one grouped launch per slot run from
    two host threads, every answer equal to search() on its own collection; config knobs when top_k is None."""
    rng = np.random.default_rng(62)
    emb = ch.HashingEmbedder()
    hs = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb, devices=[0, 0],
                           config={"topK": 4, "symbolBoost": 3.0, "searchDefaults": {"top_k": 10, "symbol_boost": 2.0},
                                   "metric": "l2", "dtype": "f32"})
    names = []
    for w, n in enumerate([700, 50, 2300, 1, 900]):
        ids, metas, docs = _chunks(n, rng)
        col = hs.create_collection(f"code_chunks__w{w}", 384)
        col.add([f"{w}-{i}" for i in ids], emb(docs), metas, docs)
        col.gpu.finalize()
        names.append(col.name)
    hs.create_collection("code_chunks__empty", 384).gpu.finalize()
    assert hs._placed == 6
    queries = ["Find the implementation of lodash chunk", "What does asciiToArray do?", "stringSize hasUnicode Parser", "zzz",
               "debounce map a", "baseFlatten", "chunk again", "isArrayLike castPath"]
    cn = [names[0], names[2], names[1], "code_chunks__nope", names[4], names[3], names[0], "code_chunks__empty"]
    for kw in (dict(top_k=8, symbol_boost=2.0), dict(top_k=3, symbol_boost=0.3, include_documents=True), dict(top_k=None, symbol_boost=None)):
        got = hs.search_many(queries, cn, **kw)
        want = [hs.search(q, collection_name=c, **kw) for q, c in zip(queries, cn)]
        assert got == want
        assert got[3] == [] and got[7] == []
    assert len(hs.search_many(queries[:1], cn[:1], top_k=None)[0]) == 4          # topK from the config
    # persistence without a host copy of the embeddings: rows are read back from the device
    from coderag_hip import store
    col = hs._cols[names[2]]
    store.save_collection(str(tmp_path), col)
    assert not hasattr(col, "_rows_parts")
    saved = np.load(os.path.join(str(tmp_path), names[2], "rows.npy"))
    assert saved.shape == (2300, 384) and np.array_equal(saved, col.gpu.rows_host())
    hs2 = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb, devices=[0, 0])
    assert hs2.search(queries[1], 8, collection_name=names[2]) == hs.search(queries[1], 8, collection_name=names[2])
    # bf16 collections persist their bit patterns
    hb = ch.HybridSearcher(db_path=str(tmp_path / "b"), embed_fn=emb, dtype="bf16")
    ids, metas, docs = _chunks(300, rng)
    cb = hb.create_collection("code_chunks__b", 384)
    cb.add(ids, emb(docs), metas, docs)
    cb.gpu.finalize()
    store.save_collection(hb.db_path, cb)
    assert np.load(os.path.join(hb.db_path, "code_chunks__b", "rows.npy")).dtype == np.uint16
    hb2 = ch.HybridSearcher(db_path=hb.db_path, embed_fn=emb, dtype="bf16")
    assert hb2.search(queries[0], 8, collection_name="code_chunks__b") == hb.search(queries[0], 8, collection_name="code_chunks__b")
    # the HTTP route validates top_k (CRAG_MAX_TOP_K) instead of failing inside the library
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from fastapi.testclient import TestClient
        client = TestClient(ch.create_app(hs))
        assert client.post("/search", json={"query": "x", "top_k": 201}, headers={"X-Workspace-Id": "w0"}).status_code == 422
        assert client.post("/search", json={"query": "x", "top_k": 0}, headers={"X-Workspace-Id": "w0"}).status_code == 422
        assert client.post("/search", json={"query": "chunk", "top_k": 200}, headers={"X-Workspace-Id": "w0"}).status_code == 200


def test_concurrent_first_access_loads_a_collection_once(tmp_path):
    from coderag_hip import store
    rng = np.random.default_rng(63)
    emb = ch.HashingEmbedder()
    hs = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb)
    ids, metas, docs = _chunks(400, rng)
    col = hs.create_collection("code_chunks__c", 384)
    col.add(ids, emb(docs), metas, docs)
    col.gpu.finalize()
    store.save_collection(hs.db_path, col)
    hs2 = ch.HybridSearcher(db_path=str(tmp_path), embed_fn=emb)
    seen, res = [], []

    def work():
        res.append(hs2.search("chunk", 5, collection_name="code_chunks__c"))
        seen.append(id(hs2._cols["code_chunks__c"]))

    ts = [threading.Thread(target=work) for _ in range(6)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert len(set(seen)) == 1 and hs2._placed == 1 and all(r == res[0] and len(r) == 5 for r in res)


def test_ingest_chunks_from_jsonl(tmp_path):
    """The step before the path (indexer/embed_ingest.py:57-132): the chunker's JSONL -> collection in batches,
    metadata sanitised, persisted, found again by a fresh searcher; records without an id get the chunker's."""
    import json
    from coderag_hip import store
    rng = np.random.default_rng(7)
    ids, metas, docs = _chunks(250, rng)
    path = tmp_path / "chunks.jsonl"
    with open(path, "w", encoding="utf-8") as f:
        for i, (cid, m, d) in enumerate(zip(ids, metas, docs)):
            m = dict(m or {"name": None, "path": f"pkg/src/f{i}.js", "kind": "Function", "start_line": i, "end_line": i + 1})
            m["tags"] = ["a", 1]                       # non-primitive -> str, None -> "" (embed_ingest.py:26-35)
            rec = {"text": d, "meta": m}
            if i % 3:
                rec["id"] = cid
            f.write(json.dumps(rec, ensure_ascii=False) + "\n")
    emb = ch.HashingEmbedder()
    hs = ch.HybridSearcher(db_path=str(tmp_path / "db"), embed_fn=emb)
    summary = store.ingest_chunks(hs, str(path), collection="code_chunks__w", batch_size=64, fresh=True)
    assert summary["input_count"] == 250 and summary["collection_count"] == 250
    q = "Find the implementation of lodash chunk"
    got = hs.search(q, top_k=8, include_documents=True, collection_name="code_chunks__w")
    assert len(got) == 8 and all(isinstance(r["metadata"].get("tags"), str) for r in got)
    recs = list(store.iter_jsonl(str(path)))
    by_id = {(r.get("id") or store.chunk_id(r["meta"])): r for r in recs}
    for r in got:
        assert r["text_full"] == by_id[r["id"]]["text"]
    hs2 = ch.HybridSearcher(db_path=str(tmp_path / "db"), embed_fn=emb)   # a fresh process would find it on disk
    assert hs2.search(q, top_k=8, include_documents=True, collection_name="code_chunks__w") == got


def test_sharded_searcher_over_rccl_world_size_1(coracle):
    """The bench's N > 1 code path (stage 1, RCCL all-gather, stage 2) with a real "nccl" process group of
    one rank: the only RCCL rehearsal a one-GPU box allows.  Results must equal the single-call search."""
    import socket
    import torch
    import torch.distributed as dist
    from coderag_hip.dist import ShardedSearcher
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    rows, queries, classes, rng = synth_case(20000, 768, "bf16", 140, seed=91)
    match = (rng.random((140, 20000)) < 0.05).astype(np.uint8)
    names, texts = names_for_match(match[:6])
    col = make_collection(rows, "bf16", "l2", names, classes, id_base=5000)
    q = torch.from_numpy(queries.view(np.int16)).cuda().view(torch.bfloat16)
    hashes = [col.symbol_hashes(t) for t in texts] + [np.zeros(0, np.uint64)] * (140 - len(texts))
    off = np.zeros(141, np.int32)
    np.cumsum([len(h) for h in hashes], out=off[1:])
    qh = torch.from_numpy(np.concatenate(hashes).view(np.int64)).cuda()
    qo = torch.from_numpy(off).cuda()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        got = ShardedSearcher(col, 0, 1).search_device(q, 10, 2.0, qh, qo)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
    want = col.search_device(q, 10, 2.0, qh, qo)
    for a, b in zip(got, want):
        assert torch.equal(a, b)


def test_library_first_then_torch_share_one_runtime():
    """A fresh process that searches through the library BEFORE torch touches the GPU: torch.cuda must still
    come up (one HIP runtime in the process, coderag_hip/_lib.py) and the device-buffer entry point must
    agree with the host-buffer one."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys\n"
        "for p in (%r, %r): sys.path.insert(0, p)\n"
        "import numpy as np\n"
        "import coderag_hip as ch\n"
        "from oracle import oracle as O\n"
        "rows = O.synth_unit_rows(3000, 384, 5, 'f32'); qs = O.synth_unit_rows(3, 384, 6, 'f32')\n"
        "col = ch.GpuCollection(384, 'f32', 'l2'); col.add(rows); col.finalize()\n"
        "ids, sc, di, cn = col.search(qs, 5, 2.0)\n"
        "import torch\n"
        "q = torch.from_numpy(qs).cuda()\n"
        "ids2, sc2, di2, cn2 = col.search_device(q, 5, 2.0)\n"
        "assert np.array_equal(ids, ids2.cpu().numpy()) and np.array_equal(sc, sc2.cpu().numpy())\n"
        "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
        "assert len(libs) == 1, libs\n"
        "print('ok')\n" % (root, os.path.join(root, "code-rag_amd")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


def _fuzz_cases(count=72, seed=20261110):
    rng = np.random.default_rng(seed)
    cases = []
    for i in range(count):
        dtype = "bf16" if rng.random() < 0.6 else "f32"
        d = int(rng.choice([int(rng.integers(1, 1025)), 128, 192, 256, 384, 512, 768, 1024]))
        nq = int(rng.choice([1, 2, 3, 8, 9, 24, 25, 32, 33, 64, 128, 129, 200, 256, 257, 300]))
        top_k = int(rng.choice([1, 2, 5, 8, 10, 10, 10, 37, 100, 200]))
        n = int(rng.choice([1, 31, 32, 33, 1000, 4096, 8191, 8192, 8193, 16385, 20000, 40000]))
        while n * nq * d > 1.5e9:
            n //= 2
        metric = "l2" if rng.random() < 0.7 else "ip"
        spread = float(rng.choice([0.0, 0.0, 0.5, 2.0]))  # log-normal spread of the row norms (0: unit rows)
        cases.append((i, dtype, d, n, nq, top_k, metric, spread))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "%02d-%s-d%d-n%d-q%d-k%d-%s-s%g" % c)
def test_seeded_fuzz_auto_path(coracle, case):
    """Seeded random shapes through the AUTO dispatch (whatever kernel the library picks), rows of widely
    different norms included (the L2 score leans on the stored |x|^2, the certificate on max |x|)."""
    i, dtype, d, n, nq, top_k, metric, spread = case
    rng = np.random.default_rng(1000 + i)
    rows = O.synth_unit_rows(n, d, 300 + i, "f32")
    if spread:
        rows = (rows * np.exp(rng.normal(0.0, spread, size=(n, 1))).astype(np.float32)).astype(np.float32)
        if n > 4:
            rows[int(rng.integers(0, n))] = 0.0  # an all-zero row
    queries = O.synth_unit_rows(nq, d, 600 + i, "f32") * np.float32(1.0 if not spread else 3.0)
    if dtype == "bf16":
        rows, queries = O.f32_to_bf16_bits(rows), O.f32_to_bf16_bits(queries)
    classes = rng.choice(np.array([0, 1, 2], np.uint8), size=n, p=[0.70, 0.05, 0.25]).astype(np.uint8)
    m = min(nq, 4)
    match = np.zeros((nq, n), np.uint8)
    match[:m] = rng.random((m, n)) < 0.08
    names, texts = names_for_match(match[:m])
    texts = texts + ["plain query"] * (nq - m)
    col = make_collection(rows, dtype, metric, names, classes)
    sb = float(rng.choice([2.0, 2.0, 0.3, 1.0]))
    check_against_oracle(coracle, col, rows, queries, classes, match, top_k, sb, metric, texts)


def test_error_codes_through_the_c_abi():
    """Error behaviour of the boundary (include/coderag_hip.h): a negative CRAG_E_* code plus a message, no
    exception across the ABI, and the handle keeps working afterwards."""
    import ctypes as C
    lib = ch.load()
    E_INVALID, E_STATE, E_LIMIT = -1, -3, -4
    vp = C.c_void_p

    def create(dim, dtype=0, metric=0):
        h = vp()
        return lib.crag_index_create(dim, dtype, metric, 0, C.byref(h)), h

    assert create(0)[0] in (E_INVALID, E_LIMIT) and lib.crag_last_error()
    assert create(4096, 0)[0] == E_LIMIT        # 16 KiB rows: beyond the kernels
    assert create(384, 7)[0] == E_INVALID       # no such dtype
    assert create(384, 0, 9)[0] == E_INVALID    # no such metric
    rc, h = create(384)
    assert rc == 0
    rows = O.synth_unit_rows(500, 384, 3, "f32")
    q = O.synth_unit_rows(2, 384, 4, "f32")
    ids = np.empty((2, 10), np.int64); sc = np.empty((2, 10)); di = np.empty((2, 10), np.float32); cn = np.empty(2, np.int32)
    qt = b"aa"; off = np.array([0, 1, 2], np.int64)

    def search(hh, b, k):
        return lib.crag_search(hh, q.ctypes.data, b, k, 2.0, qt, off.ctypes.data, ids.ctypes.data, sc.ctypes.data,
                               di.ctypes.data, cn.ctypes.data)

    assert lib.crag_index_add(h, rows.ctypes.data, 500, None, None, None) == 0
    assert search(h, 2, 10) == E_STATE and b"finalize" in lib.crag_last_error()   # search before finalize
    assert lib.crag_index_finalize(h) == 0
    assert search(h, 2, 201) == E_LIMIT                                            # CRAG_MAX_TOP_K = 200
    assert search(h, 0, 10) == E_INVALID and search(h, 2, 0) == E_INVALID
    assert search(None, 2, 10) == E_INVALID
    assert search(h, 2, 10) == 0 and list(cn) == [10, 10]                          # the handle still works
    # collections that differ in dim cannot share one grouped call
    rc, h2 = create(128)
    assert rc == 0
    r2 = O.synth_unit_rows(100, 128, 5, "f32")
    assert lib.crag_index_add(h2, r2.ctypes.data, 100, None, None, None) == 0 and lib.crag_index_finalize(h2) == 0
    hs = (vp * 2)(h, h2)
    assert lib.crag_search_multi(hs, q.ctypes.data, 2, 10, 2.0, qt, off.ctypes.data, ids.ctypes.data, sc.ctypes.data,
                                 di.ctypes.data, cn.ctypes.data) == E_INVALID
    # a collection attached to caller memory cannot grow
    import torch
    t = torch.from_numpy(rows).cuda()
    rc, h3 = create(384)
    assert rc == 0 and lib.crag_index_attach_device(h3, t.data_ptr(), 500, None, None) == 0
    assert lib.crag_index_add(h3, rows.ctypes.data, 10, None, None, None) == E_STATE
    for hh in (h, h2, h3):
        assert lib.crag_index_destroy(hh) == 0


def test_hashed_host_entry_point_equals_the_text_one():
    """crag_search_multi_hashed (the symbol side of a query as the hash list the caller already holds) returns what
    crag_search_multi returns for the query texts -- also for a text that names more chunks than symbol_hashes()'s
    first buffer holds (400 > 256), and for queries without any match."""
    n, d = 3000, 384
    rows = O.synth_unit_rows(n, d, 91, "f32")
    names = [f"sym{i:04d}x" if i < 600 else "" for i in range(n)]
    col = ch.GpuCollection(d, "f32", "l2")
    col.add(rows, names=names, path_class=(np.arange(n) % 3).astype(np.uint8))
    col.finalize()
    queries = O.synth_unit_rows(5, d, 92, "f32")
    texts = ["call " + " ".join(f"sym{i:04d}x" for i in range(400)), "only sym0007x here", "nothing", "", "SYM0599X upper"]
    hashes = [col.symbol_hashes(t) for t in texts]
    assert len(hashes[0]) == 400 and len(hashes[1]) == 1 and len(hashes[2]) == 0 and len(hashes[4]) == 1
    want = ch.search_multi([col] * 5, queries, 10, 2.0, query_texts=texts)
    got = ch.search_multi([col] * 5, queries, 10, 2.0, query_hashes=hashes)
    single = col.search(queries, 10, 2.0, query_texts=texts)
    for a, b, c in zip(want, got, single):
        assert np.array_equal(a, b) and np.array_equal(a, c)
    plain = ch.search_multi([col] * 5, queries, 10, 2.0)
    assert not np.array_equal(plain[1], want[1])     # the boost did something
    col.close()


def test_concurrent_searches_from_several_threads_are_exact(coracle):
    """Four host threads at once on two collections, host- and device-buffer entry points, batch sizes on the GEMV, the
    two-block cascade and the tiled scan, a duplicated vector so that some calls take a fallback tier: every answer
    equals the oracle's (the workspace pool, pinned staging and candidate buffers are per call).  The long form is
    tools/fuzz_threads.py."""
    import threading
    import torch
    cols, data = [], []
    for dtype, d, n in (("f32", 384, 9000), ("bf16", 768, 12000)):
        rows = O.synth_unit_rows(n, d, 7 + d, dtype)
        rows[100:140] = rows[99]
        classes = (np.arange(n) % 3).astype(np.uint8)
        queries = O.synth_unit_rows(80, d, 8 + d, dtype)
        queries[5] = rows[99]
        col = make_collection(rows, dtype, "l2", None, classes)
        want = coracle.search_numeric(rows, queries, classes, None, 10, 2.0, "l2")
        qdev = torch.from_numpy(queries).cuda() if dtype == "f32" else torch.from_numpy(queries.view(np.int16)).cuda().view(torch.bfloat16)
        cols.append(col)
        data.append((to_f32(queries), qdev, want))
    errors = []

    def client(k):
        rng = np.random.default_rng(100 + k)
        stream = torch.cuda.Stream()
        with torch.cuda.stream(stream):
            for it in range(40):
                c = int(rng.integers(0, 2))
                qf, qdev, want = data[c]
                B = int(rng.choice([1, 2, 8, 9, 20, 64]))
                lo = int(rng.integers(0, 80 - B + 1))
                if rng.random() < 0.5:
                    ids, sc, di, cn = cols[c].search(qf[lo:lo + B], 10, 2.0)
                else:
                    res = cols[c].search_device(qdev[lo:lo + B].contiguous(), 10, 2.0)
                    stream.synchronize()
                    ids, sc, di, cn = (t.cpu().numpy() for t in res)
                for b in range(B):
                    wi, ws, wd = want[lo + b]
                    kk = len(wi)
                    if not (int(cn[b]) == kk and np.array_equal(ids[b, :kk], wi) and np.array_equal(sc[b, :kk], ws)
                            and np.array_equal(di[b, :kk].view(np.uint32), wd.view(np.uint32))):
                        errors.append((k, it, c, B, lo, b))
                        return

    thr = [threading.Thread(target=client, args=(k,)) for k in range(4)]
    for t in thr:
        t.start()
    for t in thr:
        t.join()
    assert not errors, errors[:3]
    assert sum(c.stats()["fallback"] for c in cols) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,d,nq,path", [("bf16", 768, 40, "mfma"), ("f32", 384, 3, "gemv"), ("bf16", 1024, 9, "mfma"), ("f32", 384, 40, "mfma"),
                                              ("bf16", 768, 300, "mfma")])
def test_non_finite_queries_get_no_results_and_leave_the_others_exact(coracle, dtype, d, nq, path):
    """A query with a NaN or an infinite component has no nearest neighbours: count 0, and the other queries of the
    batch are answered exactly as without it (no fallback storm, no fault).  Not reference behaviour: undefined there."""
    rows, queries, classes, _ = synth_case(20000, d, dtype, nq, seed=61)
    col = make_collection(rows, dtype, "l2", classes=classes)
    col.set_path(path)
    bad = queries.copy()
    bad[0, 5] = np.uint16(0x7fc0) if dtype == "bf16" else np.float32("nan")
    bad[nq - 1, d - 1] = np.uint16(0xff80) if dtype == "bf16" else np.float32("-inf")
    ids, sc, di, cn = col.search(to_f32(bad), 10, 2.0, query_texts=[""] * nq)
    assert cn[0] == 0 and cn[nq - 1] == 0
    want = coracle.search_numeric(rows, queries, classes, None, 10, 2.0, "l2")
    for b in range(1, nq - 1):
        wi, ws, wd = want[b]
        n = len(wi)
        assert int(cn[b]) == n
        assert np.array_equal(ids[b, :n], wi) and np.array_equal(sc[b, :n], ws)
        assert np.array_equal(di[b, :n].view(np.uint32), wd.view(np.uint32))
    assert col.stats()["exact_scan"] == 0
    col.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,d", [("bf16", 768), ("f32", 384)])
@pytest.mark.parametrize("what", ["nan", "-nan", "inf"])
def test_a_collection_with_a_non_finite_row_is_rejected_at_finalize(dtype, d, what):
    rows, _, classes, _ = synth_case(3000, d, dtype, 1, seed=62)
    rows = rows.copy()
    if dtype == "bf16":
        rows[1234, 7] = {"nan": 0x7fc0, "-nan": 0xffc0, "inf": 0x7f80}[what]
    else:
        rows.view(np.uint32)[1234, 7] = {"nan": 0x7fc00000, "-nan": 0xffc00000, "inf": 0x7f800000}[what]
    with pytest.raises(ch.CragError, match="NaN or an infinite"):
        make_collection(rows, dtype, "l2", classes=classes)


@pytest.mark.gpu
def test_merge_treats_nan_distance_entries_as_padding():
    """crag_merge_rerank_device takes entries from its caller (the all-gather of the shards): an entry whose distance
    is a NaN -- either sign -- must read as padding, not rank among the real ones or leave a gap in the ranking."""
    import torch
    from coderag_hip.collection import merge_rerank_device
    entry = np.dtype([("dist", "<f4"), ("flags", "<u4"), ("id", "<i8")])
    G, B, top_k = 3, 2, 4
    kc = 2 * top_k
    e = np.zeros((G, B, kc), entry)
    e["dist"], e["id"] = np.inf, 0x7fffffffffffffff           # all padding
    real = [(0.25, 7), (0.5, 3), (1.0, 11), (1.0, 5), (3.0, 2)]
    for n, (d, i) in enumerate(real):
        e[n % G, 0, n // G] = (d, 0, i)
    e[0, 0, 5] = (np.float32("nan"), 0, 100)                  # NaN entries between real ones and padding
    e[1, 0, 4]["dist"], e[1, 0, 4]["id"] = -np.float32("nan"), 101
    e[2, 1, 0] = (np.float32("nan"), 0, 102)                  # a query with nothing but a NaN entry
    t = torch.from_numpy(e.view(np.uint8).reshape(G, B, kc, 16).copy()).cuda()
    ids, sc, di, cn = merge_rerank_device(0, t, G, B, top_k, 2.0)
    torch.cuda.synchronize()
    ids, sc, di, cn = ids.cpu().numpy(), sc.cpu().numpy(), di.cpu().numpy(), cn.cpu().numpy()
    assert cn.tolist() == [4, 0]
    assert ids[0].tolist() == [7, 3, 5, 11]                   # (distance, id) order; base score 1 / (1 + d)
    assert np.allclose(sc[0], [0.8, 1 / 1.5, 0.5, 0.5]) and (ids[1] == -1).all()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,d,nq,path", [("bf16", 768, 40, "mfma"), ("f32", 384, 3, "gemv"), ("bf16", 1024, 9, "mfma"), ("f32", 384, 40, "mfma")])
def test_queries_whose_scores_overflow_float32_go_through_every_tier_without_harm(coracle, dtype, d, nq, path):
    """A query with a component near the float32 maximum is finite (its |q|^2 fits float64), but every scan score is
    +-inf or NaN and every float32 distance is +inf: it walks through the certificate, the wide re-score, the certified
    re-scan (threshold +inf: every row is collected) and the exact scan, and ends with no result (1 / (1 + inf) = 0 is
    not > 0.01).  The other queries of the batch stay exact."""
    rows, queries, classes, _ = synth_case(20000, d, dtype, nq, seed=63)
    col = make_collection(rows, dtype, "l2", classes=classes)
    col.set_path(path)
    bad = queries.copy()
    huge = np.uint16(0x7f7f) if dtype == "bf16" else np.float32(3.0e38)
    for b in (1, nq - 1):
        bad[b, 3] = huge
    ids, sc, di, cn = col.search(to_f32(bad), 10, 2.0, query_texts=[""] * nq)
    assert cn[1] == 0 and cn[nq - 1] == 0
    want = coracle.search_numeric(rows, queries, classes, None, 10, 2.0, "l2")
    for b in range(nq):
        if b in (1, nq - 1):
            continue
        wi, ws, wd = want[b]
        n = len(wi)
        assert int(cn[b]) == n
        assert np.array_equal(ids[b, :n], wi) and np.array_equal(sc[b, :n], ws)
        assert np.array_equal(di[b, :n].view(np.uint32), wd.view(np.uint32))
    col.close()
