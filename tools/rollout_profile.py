"""Eager no-grad rollout steps at a given batch, for `rocprofv3 --kernel-trace --stats` (per-kernel view of
the unrolled-inference step).  Usage: python tools/rollout_profile.py [--batch 1] [--steps 40]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from transformerbasednavierstokesolver_amd import harness, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    args = ap.parse_args()
    cfg = synth.NS_CONFIG
    model = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=0), "cuda").eval()
    pos, a, _ = synth.ns_batch(args.batch, seed=5)
    x, fx = torch.from_numpy(pos).cuda(), torch.from_numpy(a).cuda()
    harness.rollout(model, x, fx, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    harness.rollout(model, x, fx, args.steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"eager rollout B={args.batch}: {args.steps / dt:.1f} steps/s ({1e3 * dt / args.steps:.3f} ms/step)")


if __name__ == "__main__":
    main()
