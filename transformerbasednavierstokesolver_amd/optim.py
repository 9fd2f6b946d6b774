"""Fused multi-tensor AdamW over one flat fp32 bucket (replaces the
`torch.optim.AdamW(...).step()` + `clip_grad_norm_` of exp_ns.py:117,213-217).

Parameters, gradients and both moments live in four flat buffers (parameters and gradients are views
into them, in the same order as the DDP gradient bucket, which IS this optimizer's gradient buffer),
so one kernel launch updates the whole model and the all-reduce, the norm and the update all touch the
same contiguous memory.  It is a `torch.optim.Optimizer`, so `OneCycleLR` drives `lr` and `betas[0]`
exactly as in the reference.  Parameters that never receive a gradient (`placeholder`) stay out of the
bucket and are left untouched, like `torch.optim.AdamW` does for `grad is None`.
"""
from __future__ import annotations

import torch

from . import ops
from .ddp import FlatGradSync


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise NotImplementedError("FusedAdamW supports a single parameter group")
        self.max_grad_norm = max_grad_norm
        self.sync = FlatGradSync(self.param_groups[0]["params"])     # flat gradient bucket (+ all-reduce)
        self.flat_p = self.exp_avg = self.exp_avg_sq = None
        self.steps = 0

    def _build(self):
        if self.sync.flat is None:
            self.sync._build()
        active = self.sync.active
        self.flat_p = torch.empty_like(self.sync.flat)
        off = 0
        for p in active:
            n = p.numel()
            view = self.flat_p[off:off + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            off += n
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)

    def zero_grad(self, set_to_none=False):
        if self.sync.flat is not None:
            self.sync.flat.zero_()          # keep the views alive
        else:
            super().zero_grad(set_to_none=False)

    @torch.no_grad()
    def step(self, closure=None):
        if self.flat_p is None:
            self._build()
        g = self.param_groups[0]
        self.steps += 1
        gn = ops.sumsq(self.sync.flat) if self.max_grad_norm is not None else None
        ops.adamw_step(self.flat_p, self.sync.flat, self.exp_avg, self.exp_avg_sq, float(g["lr"]),
                       float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                       self.steps, gn, float(self.max_grad_norm or 0.0))
        return None
