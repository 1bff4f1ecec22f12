import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "code-rag_amd"))
import numpy as np, torch
import coderag_hip as ch
from oracle import oracle as O
for dtype, d, nq, path in (("f32",384,1,"gemv"),("f32",384,3,"gemv"),("bf16",768,1,"gemv"),("bf16",768,8,"gemv"),("bf16",768,20,"mfma"),("bf16",1024,2,"gemv"),("f32",100,2,"gemv")):
    n = 700
    rows = O.synth_unit_rows(n, d, 1, dtype); qs = O.synth_unit_rows(nq, d, 2, dtype)
    col = ch.GpuCollection(d, dtype); col.add(rows); col.finalize()
    if dtype == "f32": q = torch.from_numpy(qs).cuda()
    else: q = torch.from_numpy(qs.view(np.int16)).cuda().view(torch.bfloat16)
    s = col.debug_dense_scores(q, path).cpu().numpy()
    X = O.as_f64(rows); Q = O.as_f64(qs)
    ref = (X*X).sum(1)[None,:] - 2*Q@X.T
    err = np.abs(s-ref)
    print(dtype, d, nq, path, "max err", err.max(), "argmax", np.unravel_index(err.argmax(), err.shape), "bad rows", int((err>1e-3).sum()))
    if err.max() > 1e-3:
        b = 0
        bad = np.where(err[b] > 1e-3)[0]
        print("  first bad rows", bad[:20], "got", s[b][bad[:5]], "want", ref[b][bad[:5]])
        # does got match some other row?
        for r in bad[:5]:
            j = np.argmin(np.abs(ref[b]-s[b][r])); print("   row", r, "looks like row", j, abs(ref[b][j]-s[b][r]))
