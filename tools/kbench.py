#!/usr/bin/env python
"""Per-stage micro-benchmark at the bench shapes (B=32, N=4096, C=256, 8 heads, M=64): times each
libpa2d stage alone with events on the launch stream and prints achieved TFLOP/s or GB/s.
Usage: python tools/kbench.py [--only conv_fwd,linear_fwd,...] [--iters 10] [--B 32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformerbasednavierstokesolver_amd import ops  # noqa: E402


def timeit(fn, iters):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--C", type=int, default=256)
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--engine", default=None, help="f32 | split | bf16 (default: PA2D_GEMM / split)")
    ap.add_argument("--H", type=int, default=64)
    ap.add_argument("--W", type=int, default=64)
    ap.add_argument("--heads", type=int, default=8)
    args = ap.parse_args()
    E = ops.resolve_engine(args.engine)
    print(f"engine {E}  B={args.B} H={args.H} W={args.W} C={args.C} M={args.M}", flush=True)
    only = set(filter(None, args.only.split(",")))
    dev = "cuda:0"
    B, H, W, C, heads, M = args.B, args.H, args.W, args.C, args.heads, args.M
    N, D = H * W, C // heads
    R = B * N
    g = torch.Generator(device=dev).manual_seed(0)
    zeros = os.environ.get("KBENCH_ZEROS") == "1"      # all-zero operands: the same instruction stream at low switching power
    rn = (lambda *s: torch.zeros(*s, device=dev)) if zeros else (lambda *s: torch.randn(*s, device=dev, generator=g))
    xn = rn(B, N, C)
    wx, wf = rn(C, C, 3, 3) * 0.02, rn(C, C, 3, 3) * 0.02
    bx, bf = rn(C), rn(C)
    dout2 = rn(B, N, 2 * C)
    w, bias = rn(C, C) * 0.06, rn(C)
    x2d, dy2d, res = xn.view(R, C), rn(R, C), rn(R, C)
    ws, bs = rn(M, D) * 0.2, rn(M) * 0.1
    temp = torch.full((heads,), 0.5, device=dev)
    wq, wk, wv = rn(D, D) * 0.2, rn(D, D) * 0.2, rn(D, D) * 0.2
    xf = rn(B, N, 2 * C)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)

    conv_flops = 2.0 * R * 9 * C * 2 * C
    lin_flops = 2.0 * R * C * C
    tests = {}
    tests["conv_fwd"] = (lambda: ops.conv3x3x2_fwd(xn, wx, bx, wf, bf, H, W, engine=E), conv_flops, "TF")
    tests["conv_bwd"] = (lambda: ops.conv3x3x2_bwd(dout2, xn, wx, wf, H, W, engine=E), 2 * conv_flops, "TF")
    tests["conv_bwd_wonly"] = (lambda: ops.conv3x3x2_bwd(dout2, xn, wx, wf, H, W, need_dx=False, engine=E), conv_flops, "TF")
    tests["linear_fwd"] = (lambda: ops.linear_fwd(x2d, w, bias, act="gelu", want_pre=True, engine=E), lin_flops, "TF")   # MLP1
    tests["linear_plain"] = (lambda: ops.linear_fwd(x2d, w, engine=E), lin_flops, "TF")
    tests["linear_bias_res"] = (lambda: ops.linear_fwd(x2d, w, bias, res=res, engine=E), lin_flops, "TF")
    tests["linear_gelu"] = (lambda: ops.linear_fwd(x2d, w, bias, act="gelu", engine=E), lin_flops, "TF")
    tests["linear_bwd_plain"] = (lambda: ops.linear_bwd_data(dy2d, w, engine=E), lin_flops, "TF")
    tests["linear_bwd_data"] = (lambda: ops.linear_bwd_data(dy2d, w, pre=x2d, act="gelu", engine=E), lin_flops, "TF")
    tests["linear_bwd_weight"] = (lambda: ops.linear_bwd_weight(dy2d, x2d, engine=E), lin_flops, "TF")
    y, mean, rstd = ops.layernorm_fwd(x2d, gamma, beta)
    tests["ln_fwd"] = (lambda: ops.layernorm_fwd(x2d, gamma, beta), 2.0 * R * C * 4, "GB")
    tests["ln_bwd"] = (lambda: ops.layernorm_bwd(dy2d, x2d, mean, rstd, gamma, res), 4.0 * R * C * 4, "GB")
    spart, npart = ops.slice_scatter(xf, 2 * C, 0, xf, 2 * C, C, ws, bs, temp, B, N, heads, D, M)
    s, nrm, o = ops.token_attn_fwd(spart, npart, wq, wk, wv)
    tests["slice_scatter"] = (lambda: ops.slice_scatter(xf, 2 * C, 0, xf, 2 * C, C, ws, bs, temp, B, N, heads, D, M),
                              2.0 * R * C * 4, "GB")
    tests["token_fwd"] = (lambda: ops.token_attn_fwd(spart, npart, wq, wk, wv), 0, "us")
    tests["deslice"] = (lambda: ops.deslice_fwd(xf, 2 * C, 0, o, ws, bs, temp, B, N, heads, D, M), 2.0 * R * C * 4, "GB")
    dy3 = dy2d.view(B, N, C)
    dopart, _ = ops.slice_scatter(xf, 2 * C, 0, dy3, C, 0, ws, bs, temp, B, N, heads, D, M, want_norm=False)
    ds, dn, *_ = ops.token_attn_bwd(s, nrm, wq, wk, wv, dopart)
    tests["token_bwd"] = (lambda: ops.token_attn_bwd(s, nrm, wq, wk, wv, dopart), 0, "us")
    tests["slice_bwd"] = (lambda: ops.slice_bwd_points(xf, dy3, ws, bs, temp, o, ds, dn, B, N, heads, D, M),
                          5.0 * R * C * 4, "GB")
    if E in (ops.ENGINE_SPLIT, ops.ENGINE_BF16) and ops.conv_planes_mask(B, H, W, C, E) == 7:
        nt = 3 if E == ops.ENGINE_SPLIT else 1
        # the forms the model runs on the bf16 engines: LayerNorm and the slice backward write the conv's plane image
        tests["ln_fwd_planes"] = (lambda: ops.layernorm_fwd_planes(x2d, gamma, beta, E), (1.0 + nt * 0.5) * R * C * 4, "GB")
        tests["slice_bwd_planes"] = (lambda: ops.slice_bwd_points_planes(xf, dy3, ws, bs, temp, o, ds, dn, nrm, B, N, heads, D, M, E),
                                     (3.0 + nt * 1.0) * R * C * 4, "GB")
    for name, (fn, work, unit) in tests.items():
        if only and name not in only:
            continue
        ms = timeit(fn, args.iters)
        if unit == "TF":
            print(f"{name:20s} {ms:9.3f} ms  {work / ms / 1e9:8.1f} TFLOP/s  ({work / ms / 1e9 / 157.3:.3f} of fp32 MFMA peak)", flush=True)
        elif unit == "GB":
            print(f"{name:20s} {ms:9.3f} ms  {work / ms / 1e6:8.0f} GB/s algorithmic", flush=True)
        else:
            print(f"{name:20s} {ms * 1e3:9.1f} us", flush=True)


if __name__ == "__main__":
    main()
