"""Extended seeded fuzz against the C oracle (not part of the test suite; run on the GPU box):
random shapes through the auto dispatch with injected duplicates / near-duplicates (so that every fallback tier is
hit), single-collection batches and multi-collection batches; symbol names and query texts (R3), path classes (R2),
symbol_boost values on both sides of the 0.01 threshold (R4); the host-buffer entry point, and with FUZZ_DEVICE=1 the
device-buffer and the no-alloc entry points compared with it; with FUZZ_POISON=1 batches that hold non-finite or
overflowing queries (those get no results, the others must not notice).

    python tools/fuzz_more.py [cases] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    sys.path.insert(0, p)
import numpy as np
import coderag_hip as ch
from oracle import coracle, oracle as O

coracle.build()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
tot = {"fallback": 0, "rescan": 0, "exact_scan": 0}
t0 = time.time()
ONLY = int(os.environ["FUZZ_ONLY"]) if "FUZZ_ONLY" in os.environ else None  # replay one case (the draws of the others are consumed)


def make_rows(n, d, dtype, dup_mode, seed):
    rows = O.synth_unit_rows(n, d, seed, "f32")
    r = np.random.default_rng(seed + 7)
    if dup_mode == 1 and n > 50:            # a few vectors with many copies
        for _ in range(3):
            src = int(r.integers(0, n))
            where = r.choice(n, size=min(n // 3, int(r.choice([30, 200, 3000]))), replace=False)
            rows[where] = rows[src]
    elif dup_mode == 2 and n > 50:          # near-duplicates: one ulp apart in one coordinate
        src = int(r.integers(0, n))
        where = r.choice(n, size=min(n // 2, 400), replace=False)
        rows[where] = rows[src]
        for w in where:
            c = int(r.integers(0, d))
            rows[w, c] = np.nextafter(rows[w, c], np.float32(np.inf if w % 2 else -np.inf))
    elif dup_mode == 3:                     # everything identical
        rows[:] = rows[0]
    if dtype == "bf16":
        rows = O.f32_to_bf16_bits(rows)
    return rows


def names_for_match(match):
    """names and query texts that realise a bool matrix match[nq <= 6, n] (as tests/test_gpu_parity.py does)"""
    nq, n = match.shape
    pat = np.zeros(n, np.int64)
    for b in range(nq):
        pat |= match[b].astype(np.int64) << b
    names = ["" if p == 0 else f"Sym{int(p):02d}q" for p in pat]
    texts = []
    for b in range(nq):
        used = sorted({names[i].lower() for i in range(n) if (pat[i] >> b) & 1})
        texts.append("find " + " and ".join(used) + " please")
    return names, texts


DEVICE = os.environ.get("FUZZ_DEVICE") == "1"
POISON = os.environ.get("FUZZ_POISON") == "1"  # also run batches with non-finite / overflowing queries in them


def check_device_entry_points(col, queries_bits, dtype, top_k, sb, texts, got, tag):
    """crag_search_device must return what crag_search returned; crag_search_device_noalloc too, within its limits
    (at most 64 queries; CRAG_E_INEXACT = -6 where a query needs a fallback tier, CRAG_E_LIMIT = -4 beyond its scratch)"""
    import ctypes as C
    import torch
    lib = ch.load()
    q = torch.from_numpy(queries_bits).cuda() if dtype == "f32" else torch.from_numpy(queries_bits.view(np.int16)).cuda().view(torch.bfloat16)
    b = q.shape[0]
    qh = qo = None
    if texts is not None:
        hs = [col.symbol_hashes(t) for t in texts]
        off = np.zeros(b + 1, np.int32)
        np.cumsum([len(h) for h in hs], out=off[1:])
        cat = np.concatenate(hs) if off[-1] else np.zeros(1, np.uint64)
        qh, qo = torch.from_numpy(cat.view(np.int64)).cuda(), torch.from_numpy(off).cuda()
    st = torch.cuda.current_stream().cuda_stream
    for name in ("crag_search_device", "crag_search_device_noalloc"):
        if name.endswith("noalloc") and b > 64:
            continue
        out = (torch.full((b, top_k), -7, dtype=torch.int64, device="cuda"), torch.full((b, top_k), -7.0, dtype=torch.float64, device="cuda"),
               torch.full((b, top_k), -7.0, dtype=torch.float32, device="cuda"), torch.full((b,), -7, dtype=torch.int32, device="cuda"))
        rc = getattr(lib, name)(col.handle, C.c_void_p(q.data_ptr()), b, top_k, C.c_double(sb),
                                C.c_void_p(qh.data_ptr()) if qh is not None else None, C.c_void_p(qo.data_ptr()) if qo is not None else None,
                                C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()), C.c_void_p(out[2].data_ptr()),
                                C.c_void_p(out[3].data_ptr()), C.c_void_p(st))
        torch.cuda.synchronize()
        if rc in (-6, -4) and name.endswith("noalloc"):
            tot["noalloc_declined"] = tot.get("noalloc_declined", 0) + 1
            continue
        if rc != 0:
            print("ERROR", name, rc, lib.crag_last_error(), tag, flush=True)
            sys.exit(1)
        ids, sc, di, cn = (t.cpu().numpy() for t in out)
        for i in range(b):
            k = int(cn[i])
            if not (k == int(got[3][i]) and np.array_equal(ids[i, :k], got[0][i, :k]) and np.array_equal(sc[i, :k], got[1][i, :k])
                    and np.array_equal(di[i, :k].view(np.uint32), got[2][i, :k].view(np.uint32))):
                print("MISMATCH", name, tag, "query", i, flush=True)
                sys.exit(1)


def check(got, want, tag):
    ids, sc, di, cn = got
    for b, (wi, ws, wd) in enumerate(want):
        k = len(wi)
        ok = int(cn[b]) == k and np.array_equal(ids[b, :k], wi) and np.array_equal(sc[b, :k], ws) and \
            np.array_equal(di[b, :k].view(np.uint32), wd.view(np.uint32))
        if not ok:
            print("MISMATCH", tag, "query", b, "count", int(cn[b]), "want", k, flush=True)
            kk = min(k, int(cn[b]))
            bad = [j for j in range(kk) if ids[b, j] != wi[j] or sc[b, j] != ws[j] or di[b, j].view(np.uint32) != wd[j].view(np.uint32)]
            print("first differing ranks", bad[:10], flush=True)
            for j in bad[:5]:
                print(j, "got", ids[b, j], sc[b, j], di[b, j], "want", wi[j], ws[j], wd[j], flush=True)
            sys.exit(1)


for case in range(n_cases):
    dtype = "bf16" if rng.random() < 0.6 else "f32"
    d = int(rng.choice([64, 100, 128, 256, 384, 512, 768, 1000, 1024, 1536]))
    if dtype == "f32" and d > 1024:
        d = 1024
    top_k = int(rng.choice([1, 5, 10, 10, 37, 100, 200]))
    metric = "l2" if rng.random() < 0.75 else "ip"
    multi = rng.random() < 0.3
    seed = 10_000 + case
    if not multi:
        n = int(rng.choice([40, 1000, 8193, 20000, 40000]))
        nq = int(rng.choice([1, 2, 3, 9, 33, 64, 130, 257]))
        while n * nq * d > 1.2e9:
            n //= 2
        dup = int(rng.choice([0, 1, 1, 2, 3])) if metric == "l2" or rng.random() < 0.5 else 0
        skip = ONLY is not None and case != ONLY
        classes = rng.integers(0, 3, size=n).astype(np.uint8)
        pick = rng.integers(0, n, size=max(1, nq // 2)) if (dup and nq > 1) else None
        if skip:
            continue
        rows = make_rows(n, d, dtype, dup, seed)
        queries = O.synth_unit_rows(nq, d, seed + 1, "f32")
        if dup and nq > 1:
            src = O.bf16_bits_to_f32(rows[pick]) if dtype == "bf16" else rows[pick]
            queries[:len(pick)] = src
        if dtype == "bf16":
            queries = O.f32_to_bf16_bits(queries)
        r2 = np.random.default_rng(seed + 3)
        match, names, texts = None, None, None
        if r2.random() < 0.5:
            m = min(nq, 4)
            match = np.zeros((nq, n), np.uint8)
            match[:m] = r2.random((m, n)) < r2.choice([0.01, 0.1, 0.6])
            names, texts = names_for_match(match[:m])
            texts = texts + ["nothing to see"] * (nq - m)
        sb = float(r2.choice([2.0, 2.0, 1.0, 0.5, 0.004, 37.5]))   # 0.004: boosted rows fall under the 0.01 threshold
        col = ch.GpuCollection(d, dtype, metric)
        col.add(rows, path_class=classes, **({"names": names} if names is not None else {}))
        col.finalize()
        qf = O.bf16_bits_to_f32(queries) if dtype == "bf16" else queries
        want = coracle.search_numeric(rows, queries, classes, match, top_k, sb, metric)
        got = col.search(qf, top_k, sb, query_texts=texts)
        check(got, want, (case, dtype, d, n, nq, top_k, metric, dup, sb, match is not None))
        if DEVICE:
            check_device_entry_points(col, queries, dtype, top_k, sb, texts, got, (case, dtype, d, n, nq, top_k, metric, dup))
        st = col.stats()
        for k in ("fallback", "rescan", "exact_scan"):
            tot[k] += st[k]
        if POISON and r2.random() < 0.5:
            # the same batch with one or two queries replaced by vectors that are not finite (NaN, +-inf: no results) or
            # whose scores overflow float32 (a component near the float32 maximum: every tier runs, anything may come
            # back for THAT query); the other queries must come back exactly as before
            bad = qf.copy()
            where = sorted({int(nq - 1 if r2.random() < 0.5 else r2.integers(0, nq)), int(r2.integers(0, nq))})
            kinds = []
            for b in where:
                kind = str(r2.choice(["nan", "inf", "-inf", "huge"]))
                kinds.append(kind)
                bad[b, int(r2.integers(0, d))] = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf, "huge": 3.0e38}[kind]
            got2 = col.search(bad, top_k, sb, query_texts=texts)
            for b, kind in zip(where, kinds):
                if kind != "huge" and int(got2[3][b]) != 0:
                    print("MISMATCH poisoned query", b, kind, "count", int(got2[3][b]), (case, dtype, d, n, nq, top_k, metric), flush=True)
                    sys.exit(1)
            keep = [b for b in range(nq) if b not in where]
            check(tuple(g[keep] for g in got2), [want[b] for b in keep], ("poisoned", case, dtype, d, n, nq, top_k, metric, where, kinds))
            tot["poisoned"] = tot.get("poisoned", 0) + 1
        col.close()
    else:
        ncol = int(rng.integers(2, 7))
        cols, data = [], []
        skip = ONLY is not None and case != ONLY
        for c in range(ncol):
            n = int(rng.choice([5, 300, 5000, 9000, 30000]))
            dup = int(rng.choice([0, 0, 1, 3]))
            classes = rng.integers(0, 3, size=n).astype(np.uint8)
            if skip:
                continue
            rows = make_rows(n, d, dtype, dup, seed + 100 * c)
            if ONLY is not None:
                print("collection", c, "rows", n, "dup", dup, flush=True)
            col = ch.GpuCollection(d, dtype, metric)
            col.add(rows, path_class=classes)
            col.set_id_base(1_000_000 * c)
            col.finalize()
            cols.append(col)
            data.append((rows, classes))
        order = sorted(int(x) for x in rng.integers(0, ncol, size=int(rng.choice([2, 5, 17, 40]))))
        if skip:
            continue
        queries = O.synth_unit_rows(len(order), d, seed + 1, "f32")
        if dtype == "bf16":
            queries = O.f32_to_bf16_bits(queries)
        qf = O.bf16_bits_to_f32(queries) if dtype == "bf16" else queries
        got = ch.search_multi([cols[i] for i in order], qf, top_k, 2.0)
        want = []
        for b, i in enumerate(order):
            rows, classes = data[i]
            want += coracle.search_numeric(rows, queries[b:b + 1], classes, None, top_k, 2.0, metric, 1_000_000 * i)
        check(got, want, (case, "multi", dtype, d, order, top_k, metric))
        if POISON and rng.random() < 0.5:
            r3 = np.random.default_rng(seed + 5)
            bad = qf.copy()
            where = sorted({int(len(order) - 1 if r3.random() < 0.5 else r3.integers(0, len(order))), int(r3.integers(0, len(order)))})
            kinds = [str(r3.choice(["nan", "inf", "-inf", "huge"])) for _ in where]
            for b, kind in zip(where, kinds):
                bad[b, int(r3.integers(0, d))] = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf, "huge": 3.0e38}[kind]
            got2 = ch.search_multi([cols[i] for i in order], bad, top_k, 2.0)
            for b, kind in zip(where, kinds):
                if kind != "huge" and int(got2[3][b]) != 0:
                    print("MISMATCH poisoned query", b, kind, "count", int(got2[3][b]), (case, "multi", dtype, d, order, top_k, metric), flush=True)
                    sys.exit(1)
            keep = [b for b in range(len(order)) if b not in where]
            check(tuple(g[keep] for g in got2), [want[b] for b in keep], ("poisoned", case, "multi", dtype, d, order, top_k, metric, where, kinds))
            tot["poisoned"] = tot.get("poisoned", 0) + 1
        st = cols[order[0]].stats()
        for k in ("fallback", "rescan", "exact_scan"):
            tot[k] += st[k]
        for c in cols:
            c.close()
    if case % 20 == 19:
        print(f"{case + 1} cases ok, {time.time() - t0:.0f} s, fallbacks so far {tot}", flush=True)
print("ALL OK", n_cases, tot, flush=True)
