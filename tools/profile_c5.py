"""configs[4] profile driver (run under rocprofv3 --kernel-trace): 64 collections x 156,250 x 1024 bf16,
one query per collection per batch through the host-buffer multi-collection entry point."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "code-rag_amd"))
import numpy as np
import torch
import coderag_hip as ch

dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(4321)


def unit(nr, dd, dtype):
    x = torch.randn((nr, dd), generator=g, device=dev, dtype=torch.float32)
    return (x / x.norm(dim=1, keepdim=True)).to(dtype).contiguous()


ncol = int(sys.argv[1]) if len(sys.argv) > 1 else 64  # collections; the 10M rows are split evenly
nper, d5, n_sym = 10_000_000 // ncol, 1024, 50_000
sym_hash, sym_len = ch.hash_names([f"sym{i}" for i in range(n_sym)])
sym_hash_t = torch.from_numpy(sym_hash.view(np.int64)).to(dev)
cols, keep = [], []
one_alloc = len(sys.argv) > 2 and sys.argv[2] == 'one'  # every collection a slice of ONE allocation
big = torch.empty((ncol * nper, d5), dtype=torch.bfloat16, device=dev) if one_alloc else None
for c in range(ncol):
    rr = unit(nper, d5, torch.bfloat16)
    if one_alloc:
        big[c * nper:(c + 1) * nper] = rr
        rr = big[c * nper:(c + 1) * nper]
    zipf = (torch.rand((nper,), generator=g, device=dev).pow(3.0) * n_sym).long().clamp_(max=n_sym - 1)
    nh = sym_hash_t[zipf].contiguous()
    u = torch.rand((nper,), generator=g, device=dev)
    pc = torch.where(u < 0.70, 0, torch.where(u < 0.95, 2, 1)).to(torch.uint8).contiguous()
    col = ch.GpuCollection(d5, "bf16", "l2", device=0)
    col.attach_device(rr, nh, pc)
    col.set_symbols(sym_hash, sym_len)
    col.finalize()
    cols.append(col)
    keep.append((rr, nh, pc))
rng = np.random.default_rng(5)
q5 = unit(ncol, d5, torch.float32).cpu().numpy()
texts = [f"where is sym{int(rng.integers(0, n_sym))} used together with sym{int(rng.integers(0, n_sym))}?" for _ in range(ncol)]
for _ in range(2):
    ch.search_multi(cols, q5, 8, 2.0, texts)
t0 = time.perf_counter()
for _ in range(10):
    ch.search_multi(cols, q5, 8, 2.0, texts)
dt = (time.perf_counter() - t0) / 10
print(f"configs[4] with {ncol} collections x {nper} rows: {dt * 1e3:.4f} ms per {ncol}-query batch, "
      f"{ncol * nper * (d5 * 2 + 4) / dt / 1e12:.2f} TB/s over the whole call", flush=True)
