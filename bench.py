#!/usr/bin/env python
"""Benchmark of the MI355X-native Transolver hot path (driver contract: one JSON line on rank 0).

Workload at every N (BASELINE.json configs[1], weak scaling): the exp_ns.py training iteration on
64x64 Navier-Stokes — 10 teacher-forced `model()` calls, summed rel-L2 loss, one backward through
all 10 graphs, AdamW(wd=1e-5) + OneCycleLR step — for Transolver_Structured_Mesh_2D with 8 layers,
C=256, 8 heads, M=64 slices, fp32, batch 32 trajectories PER GPU, synthetic seeded fields, inputs
resident in HBM.  N>1 (launched by torch.distributed.run): batch sharded across ranks, one RCCL
all-reduce(SUM) of the flat gradient bucket per iteration.

  value     = trajectories/s = N * 32 * K / (max over ranks of the time of K iterations)
  roofline  = the dominant kernel, the implicit-GEMM 3x3 conv `gemm_kc_kernel<128,128,2,2,true,32>`
              (forward and data-gradient launches have identical FLOPs): algorithmic FLOPs per launch
              2*(B*N)*(9C)*(2C) / average launch duration measured with HIP events recorded on the
              launch stream inside the timed region; peak = 157.3 TFLOP/s (fp32 MFMA, MI355X).
  cpu_baseline = the CPU oracle (port of the reference, pinned to it by oracle/make_golden.py) timed on
              this host on ONE trajectory of the same workload (rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from transformerbasednavierstokesolver_amd import synth, harness, ops, ddp  # noqa: E402
from transformerbasednavierstokesolver_amd.optim import FusedAdamW  # noqa: E402
from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss, FusedTestLoss  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0   # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, chip-level parameters
CONV_KERNEL = "gemm_kc_kernel<128,128,2,2,true,32>"


class HipEventPool:
    """Raw hipEvent_t pairs (libamdhip64 via ctypes) handed to libpa2d, which records them on the
    launch stream right around the conv implicit-GEMM kernel."""

    def __init__(self, npairs):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.pairs = []
        for _ in range(npairs):
            a, b = ctypes.c_void_p(), ctypes.c_void_p()
            assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
            self.pairs.append((a, b))
        self.used = 0
        self.enabled = False

    def provider(self):
        if not self.enabled or self.used >= len(self.pairs):
            return (0, 0)
        a, b = self.pairs[self.used]
        self.used += 1
        return (a.value, b.value)

    def durations_ms(self):
        out = []
        for a, b in self.pairs[:self.used]:
            ms = ctypes.c_float()
            if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                out.append(ms.value)
        return out

    def close(self):
        for a, b in self.pairs:
            self.hip.hipEventDestroy(a)
            self.hip.hipEventDestroy(b)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


CPU_CALLS = 5   # of the 10 teacher-forced calls of one iteration (each call costs the same)
CPU_TRAJ = 4    # trajectories in the CPU sample


def cpu_baseline(cfg, sd, pos, a, u, threads):
    """Oracle (CPU restatement == reference algorithm) on a bounded sample of the bench workload:
    CPU_TRAJ trajectories, the first CPU_CALLS of their 10 teacher-forced model calls, forward +
    backward (sized for roughly 10-30 s on the GPU host).  The iteration is 10 calls of identical
    cost, so time(trajectory) = time(sample) / CPU_TRAJ * 10/CPU_CALLS."""
    from oracle import transolver_oracle as orc
    torch.set_num_threads(threads)
    sdo = orc.to_torch(sd, torch.float32, requires_grad=True)
    x, fx, yy = (torch.from_numpy(np.ascontiguousarray(t[:CPU_TRAJ])) for t in (pos, a, u))
    t0 = time.perf_counter()
    loss, full, pred, grads = orc.train_iteration(sdo, x, fx, yy[..., :CPU_CALLS], cfg)
    dt = time.perf_counter() - t0
    return dt, pred, float(loss)


CPU_ROLLOUT_STEPS = 4


def cpu_rollout_baseline(cfg, sd, pos, a, threads):
    """Oracle prediction-feedback loop (ns_vorticity_unrolling.py:264-286) on the CPU: one trajectory,
    CPU_ROLLOUT_STEPS steps, no grad.  Returns steps/s."""
    from oracle import transolver_oracle as orc
    torch.set_num_threads(threads)
    sdo = orc.to_torch(sd, torch.float32)
    x, fx = (torch.from_numpy(np.ascontiguousarray(t[:1])) for t in (pos, a))
    with torch.no_grad():
        t0 = time.perf_counter()
        orc.rollout(sdo, x, fx, cfg, CPU_ROLLOUT_STEPS)
        return CPU_ROLLOUT_STEPS / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+backward from one hipGraph (launch-bound small batches); disables the per-kernel HIP events")
    ap.add_argument("--fold-time", action="store_true",
                    help="run the 10 teacher-forced calls of an iteration as ONE call on 10*B windows (same loss and gradients)")
    ap.add_argument("--no-folded-leg", action="store_true", help="skip the secondary time-folded measurement")
    ap.add_argument("--no-split-leg", action="store_true", help="skip the secondary split-engine measurement")
    ap.add_argument("--torch-optim", action="store_true", help="torch.optim.AdamW + torch rel-L2 instead of the fused kernels")
    ap.add_argument("--gemm-mode", type=int, default=0,
                    help="0 = exact fp32 MFMA (the metric of record), 2 = bf16-compute mode (BASELINE configs[2] numerics)")
    args = ap.parse_args()
    from transformerbasednavierstokesolver_amd import _lib
    _lib.load().pa2d_set_gemm_mode(args.gemm_mode)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with `python -m torch.distributed.run --nproc-per-node N`")
    ndev = torch.cuda.device_count()
    if world > 1 and os.environ.get("PA2D_DIST_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(ndev, 1)       # rehearsal mode: ranks share the visible GPU(s)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # "nccl" is RCCL on ROCm.  PA2D_DIST_BACKEND=gloo exists only to rehearse the N>1 code path on a
        # box with a single GPU (several ranks sharing cuda:0); it is never used for reported numbers.
        backend = os.environ.get("PA2D_DIST_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    cfg = synth.NS_CONFIG
    B = args.batch_per_gpu
    sd = synth.synth_state_dict(cfg, seed=0)          # identical on every rank
    model = harness.build_model(cfg, sd, dev).train()
    total_steps = args.steps + args.warmup
    if args.torch_optim:
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
        loss_fn = TestLoss(size_average=False)
    else:   # fused multi-tensor AdamW + fused rel-L2 (libpa2d), same arithmetic
        opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5)
        loss_fn = FusedTestLoss(size_average=False)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=3 * max(total_steps, 2) + 6)
    log(f"rank {rank}/{world}: model built, generating {B} synthetic trajectories")
    pos, a, u = synth.ns_batch(B, seed=100 + rank)    # this rank's shard of the global batch
    x, fx, yy = (torch.from_numpy(t).to(dev) for t in (pos, a, u))
    sync = ddp.FlatGradSync(model.parameters()) if args.torch_optim else opt.sync

    # parity sample before the weights move: teacher-forced predictions with the initial weights
    # (run at the full batch so that every launch of the dominant kernel in this process has the
    # same shape and the rocprofv3 per-kernel average is comparable with the live HIP-event average)
    with torch.no_grad():
        _, _, pred_gpu0 = harness.train_iteration(model, x, fx, yy)
    pred_gpu0 = pred_gpu0[:CPU_TRAJ].cpu()

    layers, calls = cfg["n_layers"], yy.shape[-1]
    pool = HipEventPool(args.steps * layers * calls * 2 + 8)
    ops.conv_event_provider = pool.provider

    if args.graph:
        ops.conv_event_provider = None
        graphed = harness.GraphedTrainStep(model, opt, sched, x, fx, yy, loss_fn=loss_fn, fold_time=args.fold_time)

    def step():
        if args.graph:
            return graphed(x, fx, yy)
        return harness.train_step(model, opt, sched, x, fx, yy, grad_sync=sync, loss_fn=loss_fn, fold_time=args.fold_time)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i + 1}/{args.warmup} done")
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    pool.enabled = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, full = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    pool.enabled = False
    log(f"{args.steps} timed steps in {dt:.2f} s")
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    conv_ms = pool.durations_ms()
    N = cfg["H"] * cfg["W"]
    C = cfg["n_hidden"]
    conv_flops = 2.0 * (B * (calls if args.fold_time else 1) * N) * (9 * C) * (2 * C)
    roof = None
    if conv_ms:
        avg_ms = float(np.mean(conv_ms))
        achieved = conv_flops / (avg_ms * 1e-3) / 1e12
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "conv_pmc_traffic.json")
        if os.path.exists(pmc) and B == 32 and not args.fold_time and args.gemm_mode == 0:   # measured on that launch shape
            try:
                traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        # exact engine: fp32 MFMA peak.  Split engine: 6 bf16 MFMA terms per fp32 product -> fp32-equivalent peak =
        # dense bf16 peak / 6.  bf16-compute mode: dense bf16 MFMA peak.
        kernel, peak = {0: (CONV_KERNEL, PEAK_FP32_MFMA_TFLOPS),
                        1: ("gemm_kc_split_kernel<128,128,true,3,true>", round(PEAK_BF16_MFMA_TFLOPS / 6, 1)),
                        2: ("gemm_kc_split_kernel<128,128,true,1,true>", PEAK_BF16_MFMA_TFLOPS)}[args.gemm_mode]
        roof = {"bound": "mfma", "kernel": kernel, "achieved": round(achieved, 2), "peak": peak,
                "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic,
                "launches_timed": len(conv_ms), "avg_launch_ms": round(avg_ms, 4),
                "flops_per_launch": conv_flops}

    out = {
        "metric": "ns64_train_samples_per_s", "value": round(world * B * args.steps / dt, 4), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32" if args.gemm_mode == 0 else ("f32 via 6-term bf16 split (conv)" if args.gemm_mode == 1 else "bf16 MFMA compute, f32 accumulate+storage"), "data": "synthetic",
        "config": {"workload": "exp_ns.py training iteration on NS 64x64 (10 teacher-forced Transolver calls + "
                               "backward + AdamW/OneCycleLR): Transolver_Structured_Mesh_2D 8 layers, C=256, 8 heads, "
                               f"M=64 slices, fp32, batch {B}/GPU (BASELINE configs[1])",
                   "global_batch": world * B, "batch_per_gpu": B, "parallelism": f"dp{world}",
                   "model_calls_per_step": calls, "grad_allreduce_bytes": sync.nbytes,
                   "hipgraph_training_step": bool(args.graph), "time_folded_calls": bool(args.fold_time)},
        "model_call_samples_per_s": round(world * B * args.steps * calls / dt, 2),
        "final_loss_per_sample_call": round(float(loss) / B / calls, 5),
        "roofline": roof,
    }

    if not (args.fold_time or args.graph or args.no_folded_leg):
        # Secondary leg, reported next to (never instead of) `value`: the same iteration with its teacher-forced
        # calls folded into one call on calls*B windows (harness.train_iteration(fold_time=True); identical
        # loss/gradients, tests/test_gpu_optim.py).  Needs the driver loop changed, so it is not the drop-in number.
        def fstep():
            return harness.train_step(model, opt, sched, x, fx, yy, grad_sync=sync, loss_fn=loss_fn, fold_time=True)
        fstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            fstep()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        fd = time.perf_counter() - t1
        if world > 1:
            tf_ = torch.tensor([fd], device=dev, dtype=torch.float64)
            dist.all_reduce(tf_, op=dist.ReduceOp.MAX)
            fd = float(tf_.item())
        out["time_folded"] = {"value": round(world * B * args.steps / fd, 4), "unit": "samples/s",
                              "ms_per_step": round(1e3 * fd / args.steps, 2),
                              "note": "10 teacher-forced calls run as one call on 10*B windows (caller-side change)"}
        log(f"time-folded leg: {out['time_folded']['value']} samples/s")

    if args.gemm_mode == 0 and not (args.fold_time or args.graph or args.no_split_leg):
        # Secondary leg, reported next to (never instead of) `value`: the same sequential iteration with the conv
        # GEMMs on the fp32-accurate 6-term bf16-split engine (pa2d_set_gemm_mode(1); passes the same fp32 parity
        # tolerances, tests/test_gpu_model.py::test_split_engine_full_ns_model_meets_fp32_tolerances).
        lib = _lib.load()
        lib.pa2d_set_gemm_mode(1)
        try:
            def sstep():
                return harness.train_step(model, opt, sched, x, fx, yy, grad_sync=sync, loss_fn=loss_fn)
            sstep()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                sstep()
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            sd_ = time.perf_counter() - t1
        finally:
            lib.pa2d_set_gemm_mode(0)
        if world > 1:
            ts_ = torch.tensor([sd_], device=dev, dtype=torch.float64)
            dist.all_reduce(ts_, op=dist.ReduceOp.MAX)
            sd_ = float(ts_.item())
        out["fp32_split_engine"] = {"value": round(world * B * args.steps / sd_, 4), "unit": "samples/s",
                                    "ms_per_step": round(1e3 * sd_ / args.steps, 2),
                                    "note": "conv GEMMs as 6 bf16 MFMA terms of an exact hi+mid+lo operand split, fp32 "
                                            "accumulate (fp32-level accuracy); opt-in engine, not the metric of record"}
        log(f"split-engine leg: {out['fp32_split_engine']['value']} samples/s")

    if not args.no_rollout:
        # unrolled-inference steps/s (ns_vorticity_unrolling.py:264-286), hipGraph-captured step.  Replicas
        # only: every rank rolls out its own trajectories, no communication; the aggregate is
        # N * batch * 20 frames / (max over ranks of the time of 20 steps).
        model.eval()
        for bsz, tag in ((B, f"b{B}"), (1, "b1")):
            gr = harness.GraphedRollout(model, x[:bsz], fx[:bsz])
            gr.run(fx[:bsz], 2)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t1 = time.perf_counter()
            gr.run(fx[:bsz], 20)
            torch.cuda.synchronize()
            d = time.perf_counter() - t1
            if world > 1:
                td = torch.tensor([d], device=dev, dtype=torch.float64)
                dist.all_reduce(td, op=dist.ReduceOp.MAX)
                d = float(td.item())
            out[f"rollout_steps_per_s_{tag}"] = round(20 / d, 2)                 # per replica
            out[f"rollout_frames_per_s_{tag}"] = round(world * 20 * bsz / d, 2)  # aggregate over all replicas
            if rank == 0:
                log(f"rollout {tag}: {20 / d:.1f} steps/s per replica, {world * 20 * bsz / d:.0f} frames/s aggregate")
            del gr

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        threads = max(1, min(threads, 16))       # a 1-GPU box owns a 16-core share
        log(f"cpu baseline: oracle on {threads} threads ...")
        cdt, pred_cpu, closs = cpu_baseline(cfg, sd, pos, a, u, threads)
        it_s = cdt * calls / CPU_CALLS / min(CPU_TRAJ, B)
        pg = pred_gpu0[..., :CPU_CALLS].double()
        rel = float((pg - pred_cpu.double()).norm() / pred_cpu.double().norm())
        out["cpu_baseline"] = {"value": round(1.0 / it_s, 5), "unit": "samples/s", "cores": threads, "kind": "port",
                               "sample": f"{min(CPU_TRAJ, B)} trajectories, {CPU_CALLS} of the {calls} teacher-forced model "
                                         f"calls of one exp_ns iteration, forward+backward, fp32 torch CPU, "
                                         f"{threads} threads: {cdt:.1f} s measured -> {it_s:.2f} s per trajectory"}
        out["rel_l2_gpu_vs_cpu_oracle"] = rel
        if not args.no_rollout:
            rs = cpu_rollout_baseline(cfg, sd, pos, a, threads)
            out["cpu_baseline"]["rollout_steps_per_s_b1"] = round(rs, 3)
            out["cpu_baseline"]["rollout_sample"] = f"{CPU_ROLLOUT_STEPS} autoregressive steps of one trajectory, no grad"
            log(f"CPU oracle rollout: {rs:.2f} steps/s at B=1")
    pool.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
