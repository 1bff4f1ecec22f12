"""crag_search_device_noalloc (include/coderag_hip.h): the entry point that allocates nothing, called through the
C ABI and checked against crag_search_device -- ids, float64 scores, float32 distances and counts bit for bit --
on two calls in a row, the second with a smaller input, for every scan kernel the dispatch can pick.
Collected last (the file name sorts behind the other tests): it must pass alone in a fresh process and after them."""
import ctypes as C

import numpy as np
import pytest

import coderag_hip as ch
from oracle import oracle as O

pytestmark = pytest.mark.gpu
E_LIMIT, E_INEXACT = -4, -6


def _dev(bits, dtype):
    import torch
    if dtype == "f32":
        return torch.from_numpy(bits).cuda()
    return torch.from_numpy(bits.view(np.int16)).cuda().view(torch.bfloat16)


def _outs(b, top_k):
    import torch
    return (torch.full((b, top_k), -7, dtype=torch.int64, device="cuda"), torch.full((b, top_k), -7.0, dtype=torch.float64, device="cuda"),
            torch.full((b, top_k), -7.0, dtype=torch.float32, device="cuda"), torch.full((b,), -7, dtype=torch.int32, device="cuda"))


def _call(fn, col, q, top_k, sb, qh, qo, out):
    import torch
    st = torch.cuda.current_stream().cuda_stream
    return fn(col.handle, C.c_void_p(q.data_ptr()), q.shape[0], top_k, C.c_double(sb),
              C.c_void_p(qh.data_ptr()) if qh is not None else None, C.c_void_p(qo.data_ptr()) if qo is not None else None,
              C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
              C.c_void_p(out[3].data_ptr()), C.c_void_p(st))


def _collection(n, d, dtype, seed, metric="l2"):
    rows = O.synth_unit_rows(n, d, seed, dtype)
    rng = np.random.default_rng(seed + 2)
    classes = rng.choice(np.array([0, 1, 2], np.uint8), size=n, p=[0.70, 0.05, 0.25]).astype(np.uint8)
    names = [f"sym{int(i)}" if i < 40 else "" for i in rng.integers(0, 400, size=n)]
    col = ch.GpuCollection(d, dtype, metric)
    col.add(rows, names=names, path_class=classes)
    col.finalize()
    return col


def _hashes(col, texts):
    import torch
    hs = [col.symbol_hashes(t) for t in texts]
    off = np.zeros(len(texts) + 1, np.int32)
    np.cumsum([len(h) for h in hs], out=off[1:])
    cat = np.concatenate(hs) if off[-1] else np.zeros(1, np.uint64)
    return torch.from_numpy(cat.view(np.int64)).cuda(), torch.from_numpy(off).cuda()


# (dtype, dim, rows, metric, [(B, top_k) first call, (B, top_k) second call]): GEMV one query, GEMV groups (float32),
# two-block MFMA cascade incl. M > 64 and the deepest cascade, split-K MFMA (1024), float32 MFMA, padded rows
CASES = [
    ("f32", 384, 40000, "l2", [(1, 10), (1, 3)]),
    ("f32", 384, 20000, "l2", [(20, 10), (3, 10)]),
    ("bf16", 768, 60000, "l2", [(64, 100), (3, 10)]),
    ("bf16", 768, 30000, "ip", [(64, 200), (40, 10)]),
    ("bf16", 1024, 20000, "l2", [(64, 10), (2, 8)]),
    ("f32", 384, 20000, "l2", [(64, 10), (25, 100)]),
    ("bf16", 1000, 9000, "l2", [(17, 10), (1, 10)]),
    ("bf16", 2048, 3000, "l2", [(64, 10), (9, 5)]),
    ("f32", 7, 300, "l2", [(5, 200), (1, 1)]),
]


@pytest.mark.parametrize("dtype,d,n,metric,calls", CASES, ids=lambda v: str(v).replace(" ", "") if not isinstance(v, list) else "calls")
def test_noalloc_entry_point_equals_crag_search_device(dtype, d, n, metric, calls):
    import torch
    lib = ch.load()
    col = _collection(n, d, dtype, seed=500 + d, metric=metric)
    for rep, (b, top_k) in enumerate(calls):
        q = _dev(O.synth_unit_rows(b, d, 900 + rep, dtype), dtype)
        texts = [f"where is sym{(7 * i) % 40} and sym{(3 * i + 1) % 40} used?" for i in range(b)]
        qh, qo = _hashes(col, texts)
        want, got = _outs(b, top_k), _outs(b, top_k)
        assert _call(lib.crag_search_device, col, q, top_k, 2.0, qh, qo, want) == 0, lib.crag_last_error()
        assert _call(lib.crag_search_device_noalloc, col, q, top_k, 2.0, qh, qo, got) == 0, lib.crag_last_error()
        torch.cuda.synchronize()
        for a, w in zip(got, want):
            assert torch.equal(a, w), (rep, b, top_k)
        assert int(want[3].min()) >= 1
    assert col.stats()["fallback"] == 0
    col.close()


def test_noalloc_two_calls_in_a_row_leave_free_memory_unchanged():
    """After a warm-up call (code objects, kernel attributes), two more calls -- the second smaller -- must not move
    the device's free memory: nothing is allocated or freed on the device."""
    import torch
    lib = ch.load()
    col = _collection(50000, 768, "bf16", seed=71)
    outs = {b: _outs(b, 10) for b in (64, 5)}
    qs = {b: _dev(O.synth_unit_rows(b, 768, 72 + b, "bf16"), "bf16") for b in (64, 5)}
    want = {b: col.search_device(qs[b], 10, 2.0) for b in (64, 5)}
    for b in (64, 5):
        assert _call(lib.crag_search_device_noalloc, col, qs[b], 10, 2.0, None, None, outs[b]) == 0, lib.crag_last_error()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for b in (64, 5):
        assert _call(lib.crag_search_device_noalloc, col, qs[b], 10, 2.0, None, None, outs[b]) == 0, lib.crag_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free0
        for a, w in zip(outs[b], want[b]):
            assert torch.equal(a, w)
    col.close()


def test_noalloc_limits_and_inexact_are_clear_errors():
    import torch
    lib = ch.load()
    col = _collection(3000, 384, "f32", seed=81)
    q = _dev(O.synth_unit_rows(65, 384, 82, "f32"), "f32")
    out = _outs(65, 10)
    assert _call(lib.crag_search_device_noalloc, col, q, 10, 2.0, None, None, out) == E_LIMIT
    assert b"64" in lib.crag_last_error()
    assert _call(lib.crag_search_device_noalloc, col, q[:4].contiguous(), 201, 2.0, None, None, _outs(4, 201)) == E_LIMIT
    col.close()
    # every vector 3000 times: the certificate cannot hold, and the fallback needs workspace this entry point has not
    base = O.synth_unit_rows(7, 768, 3, "bf16")
    rows = np.tile(base, (3000, 1))
    col = ch.GpuCollection(768, "bf16", "l2")
    col.add(rows)
    col.finalize()
    q = _dev(np.ascontiguousarray(base[:4]), "bf16")
    assert _call(lib.crag_search_device_noalloc, col, q, 10, 2.0, None, None, _outs(4, 10)) == E_INEXACT
    assert b"crag_search_device" in lib.crag_last_error()
    # ... and the handle and the scratch array are fine afterwards
    ok, want = _outs(4, 10), _outs(4, 10)
    assert _call(lib.crag_search_device, col, q, 10, 2.0, None, None, want) == 0
    col2 = _collection(5000, 768, "bf16", seed=91)
    assert _call(lib.crag_search_device_noalloc, col2, q, 10, 2.0, None, None, ok) == 0
    assert _call(lib.crag_search_device, col2, q, 10, 2.0, None, None, want) == 0
    torch.cuda.synchronize()
    for a, w in zip(ok, want):
        assert torch.equal(a, w)
