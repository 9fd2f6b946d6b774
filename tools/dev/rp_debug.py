import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from transformerbasednavierstokesolver_amd import ops
dev = "cuda:0"
torch.manual_seed(0)
for (M, N, K) in [(32768, 256, 256), (65536, 256, 256), (33280, 128, 128), (33280, 256, 128), (33280, 128, 256)]:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * K ** -0.5; b = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev)
    ref = x.double() @ w.double().t() + b.double()
    for name, kw, want in [("plain", dict(), ref), ("res", dict(res=res), ref + res.double())]:
        y, _ = ops.linear_fwd(x, w, b, engine="split", **kw)
        err = (y.double() - want).abs()
        rows = err.amax(1)
        bad = (rows > 1e-3).nonzero().flatten()
        print(M, N, K, name, "rel", float((y.double() - want).norm() / want.norm()), "bad rows", bad.numel(),
              bad[:6].tolist(), bad[-3:].tolist() if bad.numel() else "")
        if bad.numel():
            r = int(bad[0]); cols = (err[r] > 1e-3).nonzero().flatten()
            print("   row", r, "bad cols", cols.numel(), cols[:8].tolist(), "round", r // 128, "wave", (r % 128) // 32)
