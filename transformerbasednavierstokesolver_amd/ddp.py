"""Batch-sharded data parallelism over RCCL/xGMI — something the reference does not
have at all (single process, single GPU: exp_ns.py:21,34).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).  Trajectories are
independent, so the global batch is split evenly across ranks and the ONLY collective of a training
iteration is one all-reduce of the flat gradient bucket (11.2 M params = 45 MB fp32 at C=256):
  * SUM, not mean: the loss is a batch SUM (utils/testloss.py:40 with size_average=False), so the
    summed shard gradients equal the single-process gradient at the global batch;
  * parameters that receive no gradient on this path (`placeholder`, …_2D.py:205-210) keep
    `.grad is None` and stay out of the bucket, exactly like the single-process run (AdamW then
    skips them, so no spurious weight decay).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_batch(tensors, rank, world_size):
    """Rank r takes samples [r*B/ws, (r+1)*B/ws) of every tensor (B must divide evenly)."""
    out = []
    for t in tensors:
        B = t.shape[0]
        if B % world_size:
            raise ValueError(f"global batch {B} not divisible by world size {world_size}")
        n = B // world_size
        out.append(t[rank * n:(rank + 1) * n])
    return out


class FlatGradSync:
    """Flat gradient bucket + one all-reduce(SUM) per iteration.

    Built lazily after the first backward (so it knows which parameters actually receive
    gradients); from then on every `p.grad` is a view into one contiguous buffer, autograd
    accumulates in place, and the collective moves a single tensor."""

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.flat = None
        self.active = None

    def _build(self):
        self.active = [p for p in self.params if p.grad is not None]
        total = sum(p.numel() for p in self.active)
        ref = self.active[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.active:
            n = p.numel()
            view = self.flat[off:off + n].view_as(p)
            view.copy_(p.grad)
            p.grad = view
            off += n

    def __call__(self):
        if self.flat is None:
            self._build()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)

    @property
    def nbytes(self):
        return 0 if self.flat is None else self.flat.numel() * self.flat.element_size()


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (the reference seeds nothing)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
