"""Fused multi-tensor AdamW over one flat fp32 bucket (replaces the
`torch.optim.AdamW(...).step()` + `clip_grad_norm_` of exp_ns.py:117,213-217).

Parameters, gradients and both moments live in four flat buffers built EAGERLY at construction
(parameters and gradients are views into them, in the same order as the DDP gradient bucket, which IS
this optimizer's gradient buffer), so one kernel launch updates the whole model and the all-reduce,
the norm and the update all touch the same contiguous memory — and anything that captures parameter
pointers later (weight packs, hipGraphs) sees the final storage.  It is a `torch.optim.Optimizer`, so
`OneCycleLR` drives `lr` and `betas[0]` exactly as in the reference.  Parameters that have never
received a gradient (`placeholder`) are left untouched, like `torch.optim.AdamW` does for
`grad is None`; once a parameter has received one it is stepped on every iteration, which is what the
reference's `optimizer.zero_grad()` (zero tensors, not None) gives.
"""
from __future__ import annotations

import torch

from . import ops
from .ddp import FlatGradSync


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None,
                 grad_comm_dtype=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW supports a single parameter group")
        self.max_grad_norm = max_grad_norm
        # flat gradient bucket (+ all-reduce; grad_comm_dtype=torch.bfloat16 halves its wire size)
        self.sync = FlatGradSync(self.param_groups[0]["params"], comm_dtype=grad_comm_dtype)
        self.flat_p = torch.empty_like(self.sync.flat)
        with torch.no_grad():
            self.flat_p.zero_()
            for p, off in zip(self.sync.params, self.sync.offsets):
                view = self.flat_p[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.steps = 0

    def zero_grad(self, set_to_none=False):
        """Always zeroes in place (the views must stay alive); `set_to_none` is accepted and ignored."""
        self.sync.zero_()
        self.sync.attach()

    @torch.no_grad()
    def step(self, closure=None):
        self.sync.adopt()               # no-op when `sync()` already ran; catches foreign .grad tensors otherwise
        g = self.param_groups[0]
        self.steps += 1
        gn = ops.sumsq(self.sync.flat) if self.max_grad_norm is not None else None
        for a, b in self.sync.active_ranges():
            ops.adamw_step(self.flat_p[a:b], self.sync.flat[a:b], self.exp_avg[a:b], self.exp_avg_sq[a:b],
                           float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                           float(g["weight_decay"]), self.steps, gn, float(self.max_grad_norm or 0.0))
        self.sync.release_untouched()
        return None

    # ---- checkpointing: the moments and the step count round-trip like torch.optim.AdamW's state
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(), "steps": self.steps,
                       "touched": list(self.sync.touched), "numel": self.flat_p.numel()}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        fused = state_dict.pop("fused", None)
        super().load_state_dict(state_dict)
        if fused is None:
            raise KeyError("not a FusedAdamW state_dict (no 'fused' entry)")
        if fused["numel"] != self.flat_p.numel():
            raise ValueError("optimizer state belongs to a model of a different size")
        self.exp_avg.copy_(fused["exp_avg"])
        self.exp_avg_sq.copy_(fused["exp_avg_sq"])
        self.steps = int(fused["steps"])
        self.sync.touched = [bool(t) for t in fused["touched"]]
