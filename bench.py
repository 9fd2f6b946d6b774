#!/usr/bin/env python
"""Benchmark of the MI355X-native Transolver hot path (driver contract: one JSON line on rank 0).

Workload at every N (BASELINE.json configs[1], weak scaling): the exp_ns.py training iteration on
64x64 Navier-Stokes — 10 teacher-forced `model()` calls, summed rel-L2 loss, one backward through
all 10 graphs, AdamW(wd=1e-5) + OneCycleLR step — for Transolver_Structured_Mesh_2D with 8 layers,
C=256, 8 heads, M=64 slices, fp32, batch 32 trajectories PER GPU, synthetic seeded fields, inputs
resident in HBM.  N>1: batch sharded across ranks (one process per GPU), one RCCL all-reduce(SUM) of
the flat gradient bucket per iteration.  `python bench.py --gpus N` starts its own N ranks (the parent
never touches the GPU); under `python -m torch.distributed.run` it uses the ranks it was given.

  value        = trajectories/s = N * 32 * K / (max over ranks of the time of K iterations), on the default
                 GEMM engine (fp32-accurate 3xbf16 split for the conv GEMMs, exact fp32 MFMA elsewhere; the engine
                 the `-m gpu` parity suite runs); the exact-fp32-MFMA engine is measured beside it
                 (`fp32_exact_engine`)
  roofline     = the dominant kernel (implicit-GEMM 3x3 conv, forward and data-gradient launches have identical
                 FLOPs): algorithmic FLOPs per launch 2*(B*N)*(9C)*(2C) / average launch duration measured with
                 HIP events recorded on the launch stream inside the timed region
  roofline_hbm = the slice / de-slice kernels north_star names, same method, against the 8 TB/s HBM peak
  darcy421     = BASELINE configs[4] geometry (421x421, 8 layers, C=128, M=128): exp_darcy.py iteration, with
                 its own two rooflines
  cpu_baseline = the CPU oracle (port of the reference, pinned to it by oracle/make_golden.py) on this host by
                 the BASELINE.md §3 protocol (rank 0, N=1 only)
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
from transformerbasednavierstokesolver_amd.utils.normalizer import UnitTransformer  # noqa: E402

PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA, MI355X_MICROARCH.md
PEAK_FP32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBS = 8000.0             # HBM3E, MI355X_MICROARCH.md
# dominant (conv implicit-GEMM) kernel and the MFMA peak it is priced against, per engine.  Split engine: six bf16
# MFMA terms per fp32 product -> fp32-equivalent peak = dense bf16 peak / 6.
CONV_KERNELS = {ops.ENGINE_F32: ("gemm_kc_kernel<128,128,2,2,true,32>", PEAK_FP32_MFMA_TFLOPS),
                ops.ENGINE_SPLIT: ("conv_halo_kernel<3, float, 1, true>", round(PEAK_BF16_MFMA_TFLOPS / 6, 1)),   # 16x16x32 MFMA consumers
                ops.ENGINE_BF16: ("conv_halo_kernel<1, float, 3, false>", PEAK_BF16_MFMA_TFLOPS),
                ops.ENGINE_BF16S: ("conv_halo_kernel<1, __bf16, 3, false>", PEAK_BF16_MFMA_TFLOPS)}
ENGINE_LABEL = {ops.ENGINE_F32: "exact fp32 MFMA (all GEMMs)",
                ops.ENGINE_SPLIT: "3xbf16-split/fp32-acc (conv and large-M linear GEMMs, forward / data / weight "
                                  "gradients); exact fp32 MFMA for the small GEMMs",
                ops.ENGINE_BF16: "bf16 MFMA compute, f32 accumulate+storage (all GEMMs)",
                ops.ENGINE_BF16S: "bf16 storage (activations, saved tensors, inter-kernel gradients) + bf16 MFMA, f32 "
                                  "accumulate / master weights / statistics"}
# algorithmic HBM bytes per launch of the slice-path kernels, in units of R*C*e bytes (R = B*N rows, e = bytes per stored
# activation): scatter reads x_mid + v; de-slice reads x_mid, writes y; slice backward reads x_mid,
# fx_mid, dY and writes dX, dF — as e-byte tensors (5 units), or, when the engine takes the conv operands as bf16 plane
# images (ops.conv_planes_mask == 7), as the NT-plane image the conv multiplies: 3 + 2 * NT * 2 / 4 units (split: 6,
# bf16 compute: 4), which replaces the separate split pass and its traffic
HBM_KERNELS = {"slice_scatter": 2.0, "deslice": 2.0, "slice_bwd": 5.0}


def slice_kernel_names(engine, D, M, planes):
    """The names rocprofv3 prints for the three slice-path kernels of a (engine, head dim, slices) workload."""
    mt = 1 if M <= 16 else 2 if M <= 32 else 4 if M <= 64 else 8
    if engine == ops.ENGINE_F32:
        return {"slice_scatter": f"slice_scatter_kernel<{D}, {mt}, float>", "deslice": f"deslice_kernel<{D}, {mt}, float>",
                "slice_bwd": f"slice_bwd_kernel<{D}, {mt}, float, 0>"}
    t = "__bf16" if engine == ops.ENGINE_BF16S else "float"
    pl = (3 if engine == ops.ENGINE_SPLIT else 1) if planes and engine != ops.ENGINE_BF16S else 0
    # M = 128 with D != 16: the backward stays on the fp32-MFMA kernel (pa2d_slice3_bwd.hip)
    bwd = f"slice_bwd_kernel<{D}, {mt}, {t}, {pl}>" if (mt == 8 and D != 16) else f"slice_bwd3_kernel<{D}, {mt}, {t}, {pl}>"
    return {"slice_scatter": f"scatter3_kernel<{D}, {mt}, {t}>", "deslice": f"deslice3_kernel<{D}, {mt}, {t}>", "slice_bwd": bwd}


class HipEventPool:
    """Raw hipEvent_t pairs (libamdhip64 via ctypes) handed to libpa2d, which records them on the launch stream
    right around one kernel.  `provider(kind)` is installed as ops.event_provider."""

    def __init__(self):
        self.hip = ctypes.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        self.hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
        self.free, self.used = [], {}
        self.enabled = False

    def _pair(self):
        if self.free:
            return self.free.pop()
        a, b = ctypes.c_void_p(), ctypes.c_void_p()
        assert self.hip.hipEventCreate(ctypes.byref(a)) == 0 and self.hip.hipEventCreate(ctypes.byref(b)) == 0
        return (a, b)

    def provider(self, kind):
        if not self.enabled:
            return (0, 0)
        a, b = self._pair()
        self.used.setdefault(kind, []).append((a, b))
        return (a.value, b.value)

    def drain_ms(self):
        """{kind: [ms, ...]} of everything recorded so far (call after a synchronize); recycles the events."""
        out = {}
        for kind, pairs in self.used.items():
            vals = []
            for a, b in pairs:
                ms = ctypes.c_float()
                if self.hip.hipEventElapsedTime(ctypes.byref(ms), a, b) == 0:
                    vals.append(ms.value)
            out[kind] = vals
            self.free.extend(pairs)
        self.used = {}
        return out

    def close(self):
        for a, b in self.free + [p for ps in self.used.values() for p in ps]:
            self.hip.hipEventDestroy(a)
            self.hip.hipEventDestroy(b)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_info():
    model = "unknown"
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    affinity = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = None
    try:
        import psutil
        physical = psutil.cpu_count(logical=False)
    except Exception:
        pass
    quota = None                      # cgroup CPU quota (cores): the share of the host this process may really use
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(round(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return model, affinity, physical, quota


CPU_SHARE = 16      # cores a 1-GPU box owns; more threads than the share only oversubscribes (measured: 10x slower)


def cpu_baseline(cfg, sd, pos, a, u, batch, iters, rollouts, budget_s):
    """BASELINE.md §3 protocol on the oracle (CPU restatement == the reference's algorithm): config C1 (`batch` = 8
    trajectories, fp32), 1 warm-up + `iters` timed FULL exp_ns iterations (10 teacher-forced calls, summed rel-L2,
    backward, AdamW wd=1e-5, OneCycleLR step), then `rollouts` timed 20-step prediction-feedback rollouts (B=1,
    no grad).  Returns a dict (mean and best) plus the warm-up iteration's predictions for the GPU-vs-CPU rel-L2."""
    from oracle import transolver_oracle as orc
    model, affinity, physical, quota = cpu_info()
    threads = max(1, min(affinity, quota or affinity, CPU_SHARE))
    torch.set_num_threads(threads)
    sdo = orc.to_torch(sd, torch.float32, requires_grad=True)
    live = [k for k in sdo if k != "placeholder"]
    opt = torch.optim.AdamW([sdo[k] for k in live], lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=iters + 4)
    x, fx, yy = (torch.from_numpy(np.ascontiguousarray(t[:batch])) for t in (pos, a, u))
    times, pred0, note = [], None, ""
    for it in range(iters + 1):
        if it >= 2 and (it + 1) * max(times) > budget_s:       # keep the default bench run within minutes
            note = f" (stopped after {len(times)} timed iterations: {budget_s:.0f} s budget)"
            break
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, full, pred, grads = orc.train_iteration(sdo, x, fx, yy, cfg)
        for k in live:
            sdo[k].grad = grads[k]
        opt.step()
        sched.step()
        dt = time.perf_counter() - t0
        if it == 0:
            pred0 = pred          # initial weights: comparable with the GPU parity sample
            if dt > budget_s / 2:
                times, note = [dt], f" (warm-up alone took {dt:.0f} s of the {budget_s:.0f} s budget: it is the sample)"
        else:
            times.append(dt)
        log(f"cpu oracle iteration {it}/{iters} ({'warm-up' if it == 0 else 'timed'}): {dt:.1f} s")
        del grads, loss, full
        if it == 0 and note:
            break
    sd_r = orc.to_torch(sd, torch.float32)
    rtimes = []
    with torch.no_grad():
        orc.rollout(sd_r, x[:1], fx[:1], cfg, 2)                     # warm-up
        for _ in range(rollouts):
            t0 = time.perf_counter()
            orc.rollout(sd_r, x[:1], fx[:1], cfg, 20)
            rtimes.append(time.perf_counter() - t0)
    out = {"value": round(batch / float(np.mean(times)), 5), "unit": "samples/s", "cores": threads, "kind": "port",
           "best": round(batch / float(np.min(times)), 5), "cpu_model": model, "cpu_affinity": affinity,
           "cpu_physical_cores": physical, "cpu_cgroup_quota": quota,
           "sample": f"BASELINE.md §3 protocol: config C1 (B={batch}, fp32, 8 layers, C=256, M=64), 1 warm-up + {iters} "
                     f"timed full exp_ns iterations (10 calls fwd+bwd+AdamW+OneCycle), torch CPU, {threads} threads: "
                     f"{', '.join(f'{t:.1f}' for t in times)} s{note}",
           "rollout_steps_per_s_b1": round(20.0 / float(np.mean(rtimes)), 3),
           "rollout_steps_per_s_b1_best": round(20.0 / float(np.min(rtimes)), 3),
           "rollout_sample": f"{rollouts} timed 20-step prediction-feedback rollouts of one trajectory, no grad: "
                             f"{', '.join(f'{t:.2f}' for t in rtimes)} s"}
    return out, pred0


# ---------------------------------------------------------------------------------------------- launching
def self_launch(ngpus):
    """Start `python -m torch.distributed.run --nproc-per-node N bench.py <same argv>` as a CHILD process (never an
    exec: the ranks are fresh processes, the parent only waits), relay rank 0's JSON line on stdout and return the
    children's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"self-launch: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:           # the ranks log to stderr (inherited); stdout carries rank 0's JSON line
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


class Ranks:
    """Process-group context of this rank + the barrier-bracketed timer of the driver contract."""

    def __init__(self, gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if gpus != self.world:
            raise SystemExit(f"--gpus {gpus} but WORLD_SIZE={self.world}: launch with --nproc-per-node {gpus}")
        self.backend = os.environ.get("PA2D_DIST_BACKEND", "nccl") if self.world > 1 else None
        ndev = torch.cuda.device_count()
        if self.world > 1 and self.backend == "nccl" and ndev < self.world:
            raise SystemExit(f"{self.world} ranks over RCCL need {self.world} visible GPUs, found {ndev} "
                             "(PA2D_DIST_BACKEND=gloo rehearses the N>1 code path with ranks sharing a GPU)")
        if self.world > 1 and self.backend != "nccl":
            local_rank = local_rank % max(ndev, 1)       # rehearsal mode: ranks share the visible GPU(s)
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        if self.world > 1:
            # "nccl" is RCCL on ROCm.  PA2D_DIST_BACKEND=gloo exists only to rehearse the N>1 code path on a box with
            # a single GPU (several ranks sharing cuda:0); it is never used for reported numbers.
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(self, fn, steps, before_step=None):
        """Time exactly `steps` calls of fn bracketed by barrier + synchronize; MAX over ranks."""
        self.barrier()
        t0 = time.perf_counter()
        out = None
        for i in range(steps):
            if before_step is not None:
                before_step(i)
            out = fn()
        torch.cuda.synchronize()
        own = time.perf_counter() - t0                      # this rank's time before the closing barrier
        self.barrier()
        dt = time.perf_counter() - t0
        self.last_rank_ms = None
        if self.world > 1:
            tt = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            allr = [torch.zeros(1, device=self.dev, dtype=torch.float64) for _ in range(self.world)]
            dist.all_gather(allr, torch.tensor([own], device=self.dev, dtype=torch.float64))
            self.last_rank_ms = [round(1e3 * float(t.item()) / steps, 2) for t in allr]
        return dt, out

    def close(self):
        if self.world > 1:
            dist.barrier()
            dist.destroy_process_group()


def rooflines(ms_by_kind, engine, rows, C, traffic=None, planes=False, heads=8, M=64):
    """(roofline, roofline_hbm) dicts from the live HIP-event durations of one workload (rows = B*N).  planes: the slice
    backward of this workload emits the conv's plane image (see HBM_KERNELS)."""
    roof = hbm = None
    esize = 2.0 if engine == ops.ENGINE_BF16S else 4.0        # bytes per stored activation element
    conv = ms_by_kind.get("conv") or []
    if conv:
        flops = 2.0 * rows * (9 * C) * (2 * C)
        avg = float(np.mean(conv))
        kernel, peak = CONV_KERNELS[engine]
        ach = flops / (avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic["bytes"] if traffic else None,
                "traffic_source": traffic["source"] if traffic else None,
                "launches_timed": len(conv), "avg_launch_ms": round(avg, 4), "flops_per_launch": flops,
                "peak_note": ("fp32-equivalent: dense bf16 MFMA peak / 6 terms" if engine == ops.ENGINE_SPLIT else
                              ("dense bf16 MFMA" if engine in (ops.ENGINE_BF16, ops.ENGINE_BF16S) else "fp32 MFMA"))}
    planes = planes and engine in (ops.ENGINE_SPLIT, ops.ENGINE_BF16)       # bf16 STORAGE writes plain bf16 tensors
    bwd_units = 3.0 + 2.0 * (3 if engine == ops.ENGINE_SPLIT else 1) * 2.0 / 4.0 if planes else HBM_KERNELS["slice_bwd"]
    names = slice_kernel_names(engine, C // heads, M, planes)
    kernels, tot_b, tot_bs, tot_ms = [], 0.0, 0.0, 0.0
    for kind, units in HBM_KERNELS.items():
        ms = ms_by_kind.get(kind) or []
        if not ms:
            continue
        survey_bytes = units * rows * C * esize                 # 2 / 2 / 5 x R*C*e
        nbytes = (bwd_units if kind == "slice_bwd" else units) * rows * C * esize       # what this engine's kernel moves
        avg = float(np.mean(ms))
        gbs, gbs_s = nbytes / (avg * 1e-3) / 1e9, survey_bytes / (avg * 1e-3) / 1e9
        kernels.append({"kernel": names[kind], "bytes_per_launch": nbytes, "avg_launch_ms": round(avg, 4),
                        "launches_timed": len(ms), "achieved": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4),
                        "survey_bytes_per_launch": survey_bytes, "frac_survey_bytes": round(gbs_s / PEAK_HBM_GBS, 4)})
        tot_b += nbytes * len(ms)
        tot_bs += survey_bytes * len(ms)
        tot_ms += float(np.sum(ms))
    if kernels:
        ach, ach_s = tot_b / (tot_ms * 1e-3) / 1e9, tot_bs / (tot_ms * 1e-3) / 1e9
        hbm = {"bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
               "frac": round(ach / PEAK_HBM_GBS, 4), "frac_survey_bytes": round(ach_s / PEAK_HBM_GBS, 4), "traffic": None,
               "note": "launch-weighted over the three slice-path kernels (names as rocprofv3 prints them); `frac`: bytes "
                       f"this engine's kernels move = 2 / 2 / {bwd_units:g} x R*C*{int(esize)} B"
                       + (" (the slice backward writes dX | dF as the conv's bf16 plane image)" if planes else "")
                       + "; `frac_survey_bytes`: SURVEY §8(d) bytes 2 / 2 / 5 x R*C*e",
               "kernels": kernels}
    return roof, hbm


def darcy_leg(rk, pool, engine, batch, steps):
    """BASELINE configs[4]: one exp_darcy.py iteration (single model call, decode, rel-L2 + 0.1 x derivative loss,
    backward, clip 0.1, AdamW + OneCycleLR) on 421x421, 8 layers, C=128, 8 heads, M=128, batch per GPU `batch`."""
    cfg = synth.DARCY_CONFIG
    s = cfg["H"]
    model = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=7), rk.dev, engine=engine).train()
    opt = FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=0.1)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=steps + 8)
    pos, coeff, sol = synth.darcy_batch(batch, s, seed=300 + rk.rank)
    xn, yn = UnitTransformer(torch.from_numpy(coeff)), UnitTransformer(torch.from_numpy(sol))
    x = torch.from_numpy(pos).to(rk.dev)
    fx = xn.encode(torch.from_numpy(coeff)).to(rk.dev)
    y = yn.encode(torch.from_numpy(sol)).to(rk.dev)
    yn.to(rk.dev)

    def step():
        return harness.darcy_train_step(model, opt, sched, x, fx, y, yn, 1.0 / s, s, grad_sync=opt.sync)

    step()
    step()
    pool.enabled = True
    dt, (loss, l2, _) = rk.timed(step, steps)
    pool.enabled = False
    roof, hbm = rooflines(pool.drain_ms(), engine, batch * s * s, cfg["n_hidden"],
                          planes=ops.conv_planes_mask(batch, s, s, cfg["n_hidden"], engine) == 7,
                          heads=cfg["n_head"], M=cfg["slice_num"])
    fwd_gflop = 1143.0          # forward GFLOP per sample per call at this geometry
    out = {"workload": f"exp_darcy.py iteration, Darcy 421x421 (N=177241), Transolver_Structured_Mesh_2D 8 layers, "
                       f"C=128, 8 heads, M=128 slices, batch {batch}/GPU (BASELINE configs[4] geometry), "
                       f"engine {ENGINE_LABEL[engine]}",
           "value": round(rk.world * batch * steps / dt, 3), "unit": "samples/s", "ms_per_iter": round(1e3 * dt / steps, 2),
           "steps": steps, "achieved_tflops_fwd_bwd": round(3 * fwd_gflop * batch * steps / dt / 1e3, 1),
           "final_l2_per_sample": round(float(l2) / batch, 5), "roofline": roof, "roofline_hbm": hbm}
    del model, opt
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=32)
    ap.add_argument("--engine", default=None, choices=[None, "f32", "split", "bf16", "bf16s"],
                    help="GEMM engine of the measured model (default: PA2D_GEMM or the fp32-accurate split engine)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=8, help="trajectories of the CPU protocol iteration (config C1: 8)")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--cpu-budget-s", type=float, default=330.0, help="wall-clock budget of the CPU training iterations")
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--rollout-repeats", type=int, default=7, help="timed 20-step rollouts per batch size (median reported)")
    ap.add_argument("--no-bf16-leg", action="store_true",
                    help="skip the secondary bf16-storage measurement (BASELINE configs[2] / [4] as stated)")
    ap.add_argument("--graph", action="store_true",
                    help="replay forward+backward from one hipGraph (launch-bound small batches); disables the per-kernel HIP events")
    ap.add_argument("--fold-time", action="store_true",
                    help="run the 10 teacher-forced calls of an iteration as ONE call on 10*B windows (same loss and gradients)")
    ap.add_argument("--no-folded-leg", action="store_true", help="skip the secondary time-folded measurement")
    ap.add_argument("--no-exact-leg", "--no-split-leg", dest="no_exact_leg", action="store_true",
                    help="skip the secondary exact-fp32-MFMA-engine measurement")
    ap.add_argument("--no-darcy-leg", action="store_true", help="skip the Darcy 421x421 (BASELINE configs[4]) leg")
    ap.add_argument("--darcy-batch", type=int, default=2)
    ap.add_argument("--roofline-steps", type=int, default=4,
                    help="timed steps (from the first) whose conv / slice kernels carry HIP events")
    ap.add_argument("--torch-optim", action="store_true", help="torch.optim.AdamW + torch rel-L2 instead of the fused kernels")
    ap.add_argument("--bf16-grad-wire", action="store_true",
                    help="all-reduce a bf16 copy of the gradient bucket (22.5 MB instead of 45 MB; accumulation stays fp32)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU (no torch.cuda call,
        # libpa2d not loaded) and never will — it starts N fresh ranks and relays their output.
        raise SystemExit(self_launch(args.gpus))

    rk = Ranks(args.gpus)
    world, rank, dev = rk.world, rk.rank, rk.dev
    engine = ops.resolve_engine(args.engine)
    cfg = synth.NS_CONFIG
    B = args.batch_per_gpu
    sd = synth.synth_state_dict(cfg, seed=0)          # identical on every rank
    model = harness.build_model(cfg, sd, dev, engine=engine).train()
    total_steps = args.steps + args.warmup

    def make_optim(m):
        if args.torch_optim:
            o = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
            return o, ddp.FlatGradSync(m.parameters()), TestLoss(size_average=False)
        # fused multi-tensor AdamW + fused rel-L2 (§8(f)-1)
        o = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5,
                       grad_comm_dtype=torch.bfloat16 if args.bf16_grad_wire else None)
        return o, o.sync, FusedTestLoss(size_average=False)

    opt, sync, loss_fn = make_optim(model)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=4 * max(total_steps, 2) + 12)
    log(f"rank {rank}/{world}: model built (engine {engine}), generating {B} synthetic trajectories")
    pos, a, u = synth.ns_batch(B, seed=100 + rank)    # this rank's shard of the global batch
    x, fx, yy = (torch.from_numpy(t).to(dev) for t in (pos, a, u))
    calls = yy.shape[-1]
    N, C = cfg["H"] * cfg["W"], cfg["n_hidden"]

    # parity sample before the weights move: teacher-forced predictions with the initial weights (full batch, so
    # every launch of the dominant kernel in this process has the same shape and the rocprofv3 per-kernel average
    # is comparable with the live HIP-event average)
    with torch.no_grad():
        _, _, pred_gpu0 = harness.train_iteration(model, x, fx, yy)
    pred_gpu0 = pred_gpu0[:args.cpu_batch].cpu()

    pool = HipEventPool()
    if args.graph:
        graphed = harness.GraphedTrainStep(model, opt, sched, x, fx, yy, loss_fn=loss_fn, fold_time=args.fold_time)
    else:
        ops.event_provider = pool.provider

    def step():
        if args.graph:
            return graphed(x, fx, yy)
        return harness.train_step(model, opt, sched, x, fx, yy, grad_sync=sync, loss_fn=loss_fn, fold_time=args.fold_time)

    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i + 1}/{args.warmup} done")

    def gate(i):
        pool.enabled = i < args.roofline_steps
    sync.timing = world > 1
    dt, (loss, full) = rk.timed(step, args.steps, before_step=gate)
    pool.enabled = False
    rank_ms = rk.last_rank_ms
    ar_ms = sync.drain_allreduce_ms() if world > 1 else []
    sync.timing = False
    log(f"{args.steps} timed steps in {dt:.2f} s")

    traffic = None
    pmc = os.path.join(ROOT, "profiles", "conv_pmc_traffic.json")
    if os.path.exists(pmc) and B == 32 and not args.fold_time:
        try:      # PMC passes are a separate rocprofv3 run (they cannot ride in the timed region): static, labelled
            rec = json.load(open(pmc))
            if rec.get("engine", "f32") == {0: "f32", 1: "split", 2: "bf16", 3: "bf16s"}[engine]:
                traffic = {"bytes": rec.get("hbm_bytes_per_launch"),
                           "source": "profiles/conv_pmc_traffic.json (static: rocprofv3 --pmc passes of this kernel "
                                     "and launch shape, not measured in this run)"}
        except Exception:
            traffic = None
    roof, roof_hbm = rooflines(pool.drain_ms(), engine, B * (calls if args.fold_time else 1) * N, C, traffic,
                               planes=ops.conv_planes_mask(B * (calls if args.fold_time else 1), cfg["H"], cfg["W"], C, engine) == 7,
                               heads=cfg["n_head"], M=cfg["slice_num"])

    out = {
        "metric": "ns64_train_samples_per_s", "value": round(world * B * args.steps / dt, 4), "unit": "samples/s",
        "n_gpus": world,
        "rccl_ranks": (dist.get_world_size() if world > 1 and rk.backend == "nccl" else (1 if world == 1 else 0)),
        "dist_backend": rk.backend, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * dt / args.steps, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {ops.ENGINE_F32: "f32", ops.ENGINE_SPLIT: "f32", ops.ENGINE_BF16: "bf16 MFMA compute, f32 accumulate+storage",
                  ops.ENGINE_BF16S: "bf16"}[engine], "data": "synthetic",
        "config": {"workload": "exp_ns.py training iteration on NS 64x64 (10 teacher-forced Transolver calls + "
                               "backward + AdamW/OneCycleLR): Transolver_Structured_Mesh_2D 8 layers, C=256, 8 heads, "
                               f"M=64 slices, fp32, batch {B}/GPU (BASELINE configs[1])",
                   "gemm_engine": ENGINE_LABEL[engine],
                   "global_batch": world * B, "batch_per_gpu": B, "parallelism": f"dp{world}",
                   "model_calls_per_step": calls, "grad_allreduce_bytes": sync.nbytes,
                   "hipgraph_training_step": bool(args.graph), "time_folded_calls": bool(args.fold_time)},
        "model_call_samples_per_s": round(world * B * args.steps * calls / dt, 2),
        "final_loss_per_sample_call": round(float(loss) / B / calls, 5),
        "roofline": roof, "roofline_hbm": roof_hbm,
    }
    if world > 1:
        # what an N > 1 line needs to explain itself: the collective's device time (rank 0's HIP events around
        # dist.all_reduce of the flat gradient bucket, mean over the timed steps; it runs after backward, not overlapped:
        # with ten accumulating calls per iteration a parameter's gradient is final only in the last call's backward) and
        # every rank's own step time
        out["allreduce_ms"] = round(float(np.mean(ar_ms)), 3) if ar_ms else None
        out["allreduce_gbs_per_rank"] = (round(2.0 * (world - 1) / world * sync.nbytes / (np.mean(ar_ms) * 1e-3) / 1e9, 1)
                                         if ar_ms else None)
        out["step_ms_per_rank"] = {"min": min(rank_ms), "max": max(rank_ms), "all": rank_ms} if rank_ms else None

    if not (args.fold_time or args.graph or args.no_folded_leg):
        # Secondary leg, reported next to (never instead of) `value`: the same iteration with its teacher-forced
        # calls folded into one call on calls*B windows (harness.train_iteration(fold_time=True); identical
        # loss/gradients, tests/test_gpu_optim.py).  Needs the driver loop changed, so it is not the drop-in number.
        def fstep():
            return harness.train_step(model, opt, sched, x, fx, yy, grad_sync=sync, loss_fn=loss_fn, fold_time=True)
        fstep()
        fsteps = min(args.steps, 5)
        fd, _ = rk.timed(fstep, fsteps)
        out["time_folded"] = {"value": round(world * B * fsteps / fd, 4), "unit": "samples/s",
                              "ms_per_step": round(1e3 * fd / fsteps, 2), "steps": fsteps,
                              "note": "10 teacher-forced calls run as one call on 10*B windows (caller-side change)"}
        log(f"time-folded leg: {out['time_folded']['value']} samples/s")

    if engine != ops.ENGINE_F32 and not (args.fold_time or args.graph or args.no_exact_leg):
        # Secondary leg: the same sequential iteration on the exact-fp32-MFMA engine (every GEMM on
        # v_mfma_f32_32x32x2_f32) — a second model + optimizer on the same weights and data, own engine argument.
        m2 = harness.build_model(cfg, sd, dev, engine="f32").train()
        o2, s2, l2 = make_optim(m2)

        def estep():
            return harness.train_step(m2, o2, None, x, fx, yy, grad_sync=s2, loss_fn=l2)
        estep()
        esteps = min(args.steps, 5)
        pool.enabled = True
        ed, _ = rk.timed(estep, esteps)
        pool.enabled = False
        eroof, _ = rooflines(pool.drain_ms(), ops.ENGINE_F32, B * N, C, heads=cfg["n_head"], M=cfg["slice_num"])
        out["fp32_exact_engine"] = {"value": round(world * B * esteps / ed, 4), "unit": "samples/s",
                                    "ms_per_step": round(1e3 * ed / esteps, 2), "steps": esteps, "roofline": eroof,
                                    "note": "every GEMM on v_mfma_f32_32x32x2_f32 (engine f32); same weights, data, loop"}
        log(f"exact-engine leg: {out['fp32_exact_engine']['value']} samples/s")
        del m2, o2, s2
        torch.cuda.empty_cache()

    if not args.no_rollout:
        # unrolled-inference steps/s (ns_vorticity_unrolling.py:264-286), hipGraph-captured step.  Replicas
        # only: every rank rolls out its own trajectories, no communication; the aggregate is
        # N * batch * 20 frames / (max over ranks of the time of 20 steps).
        ops.event_provider = None
        model.eval()
        for bsz, tag in ((B, f"b{B}"), (1, "b1")):
            gr = harness.GraphedRollout(model, x[:bsz], fx[:bsz])
            gr.run(fx[:bsz], 2)
            ds = sorted(rk.timed(lambda: gr.run(fx[:bsz], 20), 1)[0] for _ in range(args.rollout_repeats))
            d = ds[len(ds) // 2]                                                  # median of the repeats
            out[f"rollout_steps_per_s_{tag}"] = round(20 / d, 2)                 # per replica
            out[f"rollout_steps_per_s_{tag}_best"] = round(20 / ds[0], 2)
            out[f"rollout_frames_per_s_{tag}"] = round(world * 20 * bsz / d, 2)  # aggregate over all replicas
            if rank == 0:
                log(f"rollout {tag}: {20 / d:.1f} steps/s per replica (median of {len(ds)} 20-step rollouts, best "
                    f"{20 / ds[0]:.1f}), {world * 20 * bsz / d:.0f} frames/s aggregate")
            del gr
        out["rollout_sample"] = f"{args.rollout_repeats} timed 20-step graph-captured rollouts per batch size: median (and best)"
        model.train()

    if not args.no_darcy_leg:
        ops.event_provider = pool.provider
        torch.cuda.empty_cache()
        out["darcy421"] = darcy_leg(rk, pool, engine, args.darcy_batch, min(max(args.steps, 3), 5))
        log(f"darcy421 leg: {out['darcy421']['ms_per_iter']} ms/iter, {out['darcy421']['value']} samples/s")

    if engine != ops.ENGINE_BF16S and not (args.fold_time or args.graph or args.no_bf16_leg):
        # Secondary leg: BASELINE configs[2] / [4] AS STATED — bf16.  The same exp_ns.py iteration (and the same Darcy
        # iteration) on the bf16-STORAGE path (activations, saved tensors and inter-kernel gradients bf16 in HBM; fp32
        # master weights, statistics and accumulators; optional bf16 gradient wire): a second model + optimizer on the
        # same weights and this rank's shard, so an N > 1 run measures configs[2] next to the fp32-accurate `value`.
        ops.event_provider = pool.provider
        torch.cuda.empty_cache()
        mb = harness.build_model(cfg, sd, dev, engine="bf16s").train()
        ob = FusedAdamW(mb.parameters(), lr=1e-3, weight_decay=1e-5, grad_comm_dtype=torch.bfloat16)
        lb = FusedTestLoss(size_average=False)

        def bstep():
            return harness.train_step(mb, ob, None, x, fx, yy, grad_sync=ob.sync, loss_fn=lb)
        bstep()
        bsteps = min(args.steps, 8)
        pool.enabled = True
        ob.sync.timing = world > 1
        bd, (bloss, _) = rk.timed(bstep, bsteps)
        pool.enabled = False
        bar = ob.sync.drain_allreduce_ms() if world > 1 else []
        broof, bhbm = rooflines(pool.drain_ms(), ops.ENGINE_BF16S, B * N, C, heads=cfg["n_head"], M=cfg["slice_num"])
        leg = {"workload": f"the same exp_ns.py iteration, batch {B}/GPU, {ENGINE_LABEL[ops.ENGINE_BF16S]}; bf16 gradient "
                           "wire (BASELINE configs[2] as stated)",
               "value": round(world * B * bsteps / bd, 4), "unit": "samples/s", "ms_per_step": round(1e3 * bd / bsteps, 2),
               "steps": bsteps, "dtype": "bf16", "global_batch": world * B, "grad_allreduce_bytes": ob.sync.nbytes,
               "final_loss_per_sample_call": round(float(bloss) / B / calls, 5), "roofline": broof, "roofline_hbm": bhbm}
        if world > 1:
            leg["allreduce_ms"] = round(float(np.mean(bar)), 3) if bar else None
            leg["step_ms_per_rank"] = {"min": min(rk.last_rank_ms), "max": max(rk.last_rank_ms)} if rk.last_rank_ms else None
        del mb, ob
        torch.cuda.empty_cache()
        if not args.no_darcy_leg:
            leg["darcy421"] = darcy_leg(rk, pool, ops.ENGINE_BF16S, args.darcy_batch, min(max(args.steps, 3), 5))
        out["bf16_storage"] = leg
        log(f"bf16-storage leg: {leg['value']} samples/s"
            + (f", darcy421 {leg['darcy421']['ms_per_iter']} ms/iter" if "darcy421" in leg else ""))

    ops.event_provider = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log("cpu baseline: oracle by the BASELINE.md §3 protocol ...")
        cb = min(args.cpu_batch, B)
        rec, pred_cpu = cpu_baseline(cfg, sd, pos, a, u, cb, args.cpu_iters, 3, args.cpu_budget_s)
        out["cpu_baseline"] = rec
        out["rel_l2_gpu_vs_cpu_oracle"] = float((pred_gpu0[:cb].double() - pred_cpu.double()).norm() / pred_cpu.double().norm())
        log(f"CPU oracle: {rec['value']} samples/s, rollout {rec['rollout_steps_per_s_b1']} steps/s at B=1")
    pool.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
