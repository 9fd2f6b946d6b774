"""Callers of the hot path, restated for the MI355X build.

The reference's driver scripts execute argparse and a hard-coded Windows data path at import time,
so they cannot be imported; these functions reproduce their loops:

  train_iteration / train_step  exp_ns.py:191-218   (T/step teacher-forced model calls, summed rel-L2,
                                                    one backward, AdamW(wd=1e-5) + OneCycleLR step)
  rollout                       exp_ns.py:225-241, ns_vorticity_unrolling.py:264-286
                                (prediction-feedback loop, no grad)
  GraphedRollout                the same step captured once in a hipGraph (static buffers; the
                                `torch.cat` window shift becomes an in-place roll)
"""
from __future__ import annotations

import torch

from .model.Transolver_Structured_Mesh_2D import Model
from .utils.testloss import TestLoss


def build_model(cfg, state_dict=None, device="cuda"):
    m = Model(space_dim=cfg["space_dim"], n_layers=cfg["n_layers"], n_hidden=cfg["n_hidden"],
              dropout=cfg.get("dropout", 0.0), n_head=cfg["n_head"], Time_Input=cfg["Time_Input"],
              act=cfg.get("act", "gelu"), mlp_ratio=cfg["mlp_ratio"], fun_dim=cfg["fun_dim"],
              out_dim=cfg["out_dim"], slice_num=cfg["slice_num"], ref=cfg["ref"],
              unified_pos=cfg["unified_pos"], H=cfg["H"], W=cfg["W"])
    if state_dict is not None:
        m.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return m.to(device)


def train_iteration(model, x, fx, yy, step=1, loss_fn=None):
    """Forward part of one exp_ns mini-batch.  Returns (loss [graph attached], full_loss, pred)."""
    loss_fn = loss_fn or TestLoss(size_average=False)
    T = yy.shape[-1]
    bsz = x.shape[0]
    loss = 0
    preds = []
    for t in range(0, T, step):
        y = yy[..., t:t + step]
        im = model(x, fx=fx)
        loss = loss + loss_fn(im.reshape(bsz, -1), y.reshape(bsz, -1))
        preds.append(im)
        fx = torch.cat((fx[..., step:], y), dim=-1)          # teacher forcing with ground truth
    pred = torch.cat(preds, -1)
    with torch.no_grad():
        full = loss_fn(pred.reshape(bsz, -1), yy.reshape(bsz, -1))
    return loss, full, pred


def train_step(model, optimizer, scheduler, x, fx, yy, step=1, max_grad_norm=None, grad_sync=None,
               set_to_none=False):
    """One full exp_ns.py:191-218 iteration.  `grad_sync` (DDP): callable run between backward and
    the optimizer step (all-reduce SUM of the flat gradient bucket)."""
    loss, full, _ = train_iteration(model, x, fx, yy, step)
    optimizer.zero_grad(set_to_none=set_to_none)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    if max_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), full


@torch.no_grad()
def rollout(model, x, fx, nsteps, step=1):
    frames = []
    for _ in range(nsteps):
        im = model(x, fx=fx)
        frames.append(im)
        fx = torch.cat((fx[..., step:], im), dim=-1)
    return torch.cat(frames, -1)


class GraphedRollout:
    """One autoregressive step (model call + window shift) captured in a hipGraph and replayed.

    Static buffers: `x`, the input window `fx` [B,N,fun_dim] and the output `im`.  The window shift
    `fx = cat(fx[..., step:], im)` is done in place inside the graph, so shapes and pointers never
    change.  Replays are bit-identical to the eager loop (same kernels, same order)."""

    def __init__(self, model, x, fx, step=1, warmup=2):
        self.model, self.step = model, step
        self.x = x.clone()
        self.fx = fx.clone()
        self.graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):
                self._step_eager()
        torch.cuda.current_stream().wait_stream(side)
        self.fx.copy_(fx)
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self._step_eager()
        self.fx.copy_(fx)

    def _step_eager(self):
        self.im = self.model(self.x, fx=self.fx)
        k = self.im.shape[-1]
        shifted = self.fx[..., self.step:].clone()
        self.fx[..., :-k].copy_(shifted[..., :self.fx.shape[-1] - k])
        self.fx[..., -k:].copy_(self.im)

    @torch.no_grad()
    def run(self, fx0, nsteps):
        self.fx.copy_(fx0)
        frames = []
        for _ in range(nsteps):
            self.graph.replay()
            frames.append(self.im.clone())
        return torch.cat(frames, -1)
