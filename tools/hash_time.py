import sys, os, time
sys.path.insert(0, "code-rag_amd")
import numpy as np
import coderag_hip as ch
n_sym = 50000
sym_hash, sym_len = ch.hash_names([f"sym{i}" for i in range(n_sym)])
col = ch.GpuCollection(64, "bf16", "l2", device=0)
col.add(np.zeros((10, 64), np.float32).view(np.float32))
col.set_symbols(sym_hash, sym_len)
col.finalize()
texts = [f"where is sym{i*37} used together with sym{i*91}?" for i in range(64)]
t0 = time.perf_counter()
for _ in range(10):
    for t in texts: col.symbol_hashes(t)
print("symbol_hashes per 64 queries: %.3f ms" % ((time.perf_counter() - t0) / 10 * 1e3))
