"""Batch-sharded data parallelism over RCCL/xGMI — something the reference does not
have at all (single process, single GPU: exp_ns.py:21,34).

One process per GPU (`torch.distributed`, backend "nccl" == RCCL on ROCm).  Trajectories are
independent, so the global batch is split evenly across ranks and the ONLY collective of a training
iteration is one all-reduce of the flat gradient bucket (11.2 M params = 45 MB fp32 at C=256):
  * SUM, not mean: the loss is a batch SUM (utils/testloss.py:40 with size_average=False), so the
    summed shard gradients equal the single-process gradient at the global batch;
  * parameters that have never received a gradient on this path (`placeholder`, …_2D.py:205-210) are
    handed back to the optimizer with `.grad is None`, exactly like the single-process run (AdamW
    then skips them, so no spurious weight decay); their slot in the bucket stays zero.
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

    Built EAGERLY over every trainable parameter (fixed offsets, 16-byte aligned slots): each `p.grad`
    is a view into one contiguous buffer from construction on, so autograd — and the libpa2d weight-
    gradient kernels, which accumulate straight into these views — never copy, and the collective moves
    a single tensor.  The bucket owns `p.grad`: `__call__` re-adopts any gradient that is not its view
    (e.g. after `zero_grad(set_to_none=True)`, which makes autograd allocate fresh tensors) by copying
    it into the slot, so the reduced buffer can never be stale.  `touched[i]` (sticky) records which
    parameters have ever received a gradient — the reference's `optimizer.zero_grad()` keeps such
    gradients as zero tensors, so they are stepped from then on; the others keep `.grad = None`."""

    ALIGN = 4       # floats: every slot starts on a 16-byte boundary (vector loads of the fused kernels)

    def __init__(self, params, group=None, comm_dtype=None):
        """`comm_dtype=torch.bfloat16`: the all-reduce moves a bf16 copy of the bucket (half the xGMI payload: 22.5 MB
        instead of 45 MB at C=256); accumulation into the bucket and the optimizer stay fp32."""
        self.comm_dtype = comm_dtype
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatGradSync needs at least one trainable parameter")
        self.group = group
        ref = self.params[0]
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != ref.dtype or p.device != ref.device:
                raise ValueError("all parameters of one bucket must share dtype and device")
            self.offsets.append(off)
            off += -(-p.numel() // self.ALIGN) * self.ALIGN
        self.total = off
        self.flat = torch.zeros(off, dtype=ref.dtype, device=ref.device)
        self.views = [self.flat[o:o + p.numel()].view_as(p) for o, p in zip(self.offsets, self.params)]
        self.touched = [False] * len(self.params)
        self._hooks = []
        for i, p in enumerate(self.params):
            if p.grad is not None:                 # adopt what is already there
                self.views[i].copy_(p.grad)
                self.touched[i] = True
            p.grad = self.views[i]
            p._pa2d_slot = (self, i)       # functional.grad_targets: backward kernels accumulate straight into views[i]
            self._hooks.append(p.register_post_accumulate_grad_hook(self._mark(i)))

    def _mark(self, i):
        def hook(_p):
            self.touched[i] = True
        return hook

    def index_of(self, p):
        for i, q in enumerate(self.params):
            if q is p:
                return i
        raise KeyError("parameter is not in this bucket")

    def target(self, i):
        """Gradient buffer of parameter i for in-place accumulation by a backward kernel (functional.grad_targets):
        the slot's view, made the parameter's `.grad` first (zeroed if the parameter had no gradient yet this
        iteration, seeded with a foreign `.grad` tensor if there is one)."""
        p, v = self.params[i], self.views[i]
        g = p.grad
        if g is None:
            v.zero_()
            p.grad = v
        elif g.data_ptr() != v.data_ptr():
            v.copy_(g)
            p.grad = v
        self.touched[i] = True
        return v

    def attach(self):
        """Before backward: every slot's view is (again) the parameter's `.grad`, so gradients accumulate in place."""
        for p, v in zip(self.params, self.views):
            if p.grad is None:
                v.zero_()
                p.grad = v

    def adopt(self):
        """After backward: make the bucket hold every gradient (copy foreign tensors in, zero absent ones)."""
        for i, (p, v) in enumerate(zip(self.params, self.views)):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.shape != v.shape:
                if g.shape != v.shape:
                    raise RuntimeError("a parameter changed shape after the gradient bucket was built")
                v.copy_(g)
                p.grad = v
                self.touched[i] = True

    def release_untouched(self):
        """Parameters that never received a gradient go back to `.grad = None` (torch optimizers skip them)."""
        for i, p in enumerate(self.params):
            if not self.touched[i]:
                p.grad = None

    def active_ranges(self):
        """Contiguous [start, stop) float ranges of the flat buffer covered by touched parameters."""
        runs = []
        for i, p in enumerate(self.params):
            if not self.touched[i]:
                continue
            a = self.offsets[i]
            b = self.offsets[i + 1] if i + 1 < len(self.params) else self.total
            if runs and runs[-1][1] == a:
                runs[-1][1] = b
            else:
                runs.append([a, b])
        return [(a, b) for a, b in runs]

    def zero_(self):
        self.flat.zero_()

    def __call__(self):
        self.adopt()
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            ev = None
            if getattr(self, "timing", False) and self.flat.is_cuda:      # bench: device time of the collective (+ wire casts)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            if self.comm_dtype is not None and self.comm_dtype != self.flat.dtype:
                wire = self.flat.to(self.comm_dtype)
                dist.all_reduce(wire, op=dist.ReduceOp.SUM, group=self.group)
                self.flat.copy_(wire)
            else:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
            if ev is not None:
                ev[1].record()
                self._events.append(ev)
        self.release_untouched()

    timing = False

    @property
    def _events(self):
        if not hasattr(self, "_ev_list"):
            self._ev_list = []
        return self._ev_list

    def drain_allreduce_ms(self):
        """Device-side durations (ms) of the all-reduces issued since the last call, `timing` enabled (synchronize first)."""
        out = [a.elapsed_time(b) for a, b in self._events]
        self._ev_list = []
        return out

    @property
    def nbytes(self):
        """bytes one all-reduce moves"""
        es = self.flat.element_size() if self.comm_dtype is None else torch.empty(0, dtype=self.comm_dtype).element_size()
        return self.flat.numel() * es


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s weights (the reference seeds nothing)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src, group=group)
