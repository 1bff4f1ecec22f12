"""Concurrency stress with verification: several host threads search the same two collections at once -- host-buffer and
device-buffer entry points, single and multi-collection calls, batch sizes on every scan family -- and every answer is
compared with the C oracle's (computed once up front).  Catches races in the workspace pool, the pinned staging, the
candidate buffers and the pacing words, which the single-threaded fuzz cannot.

    python tools/fuzz_threads.py [threads] [calls per thread] [seed]
"""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "code-rag_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import coderag_hip as ch
from oracle import coracle, oracle as O

coracle.build()
n_thr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
per_thr = int(sys.argv[2]) if len(sys.argv) > 2 else 150
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1

cols, data = [], []
for dtype, d, n in (("f32", 384, 20000), ("bf16", 768, 30011), ("bf16", 1024, 9000), ("bf16", 1536, 6000)):
    rows = O.synth_unit_rows(n, d, 7 + d, dtype)
    rows[100:160] = rows[99]                       # a few duplicates: some queries take a fallback tier
    classes = (np.arange(n) % 3).astype(np.uint8)
    queries = O.synth_unit_rows(300, d, 8 + d, dtype)
    queries[5] = rows[99]
    col = ch.GpuCollection(d, dtype, "l2")
    col.add(rows, path_class=classes)
    col.finalize()
    want = {tk: coracle.search_numeric(rows, queries, classes, None, tk, 2.0, "l2") for tk in (10, 100)}
    qf = O.bf16_bits_to_f32(queries) if dtype == "bf16" else queries
    qdev = torch.from_numpy(queries).cuda() if dtype == "f32" else torch.from_numpy(queries.view(np.int16)).cuda().view(torch.bfloat16)
    cols.append(col)
    data.append((qf, qdev, want))
print("collections and oracle answers ready", flush=True)
errors, done = [], [0] * n_thr


def same(got, want_list, lo):
    ids, sc, di, cn = got
    for b in range(len(cn)):
        wi, ws, wd = want_list[lo + b]
        k = len(wi)
        if not (int(cn[b]) == k and np.array_equal(ids[b, :k], wi) and np.array_equal(sc[b, :k], ws)
                and np.array_equal(di[b, :k].view(np.uint32), wd.view(np.uint32))):
            return False
    return True


def client(k):
    rng = np.random.default_rng(seed * 1000 + k)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for it in range(per_thr):
            c = int(rng.integers(0, len(cols)))
            qf, qdev, want = data[c]
            B = int(rng.choice([1, 1, 2, 5, 8, 9, 20, 64, 130, 257]))
            lo = int(rng.integers(0, 300 - B + 1))
            tk = int(rng.choice([10, 10, 100]))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                got = cols[c].search(qf[lo:lo + B], tk, 2.0)
            elif mode == 1:
                res = cols[c].search_device(qdev[lo:lo + B].contiguous(), tk, 2.0)
                stream.synchronize()
                got = tuple(t.cpu().numpy() for t in res)
            else:
                got = ch.search_multi([cols[c]] * B, qf[lo:lo + B], tk, 2.0)
            if not same(got, want[tk], lo):
                errors.append((k, it, c, B, lo, tk, mode))
                return
            done[k] += 1


t0 = time.time()
thr = [threading.Thread(target=client, args=(k,)) for k in range(n_thr)]
for t in thr:
    t.start()
for t in thr:
    t.join()
st = [c.stats() for c in cols]
print(f"{sum(done)} calls from {n_thr} threads in {time.time() - t0:.1f} s; fallbacks {[s['fallback'] for s in st]}; "
      f"mfma/gemv batches {[(s['mfma_batches'], s['gemv_batches']) for s in st]}", flush=True)
if errors:
    print("MISMATCH (thread, call, collection, B, first query, top_k, entry point):", errors[:5], flush=True)
    sys.exit(1)
print("ALL OK", flush=True)
