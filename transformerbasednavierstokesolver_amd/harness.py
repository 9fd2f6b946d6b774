"""Callers of the hot path, restated for the MI355X build.

The reference's driver scripts execute argparse and a hard-coded Windows data path at import time,
so they cannot be imported; these functions reproduce their loops:

  train_iteration / train_step  exp_ns.py:191-218   (T/step teacher-forced model calls, summed rel-L2,
                                                    one backward, AdamW(wd=1e-5) + OneCycleLR step)
  rollout                       exp_ns.py:225-241, ns_vorticity_unrolling.py:264-286
                                (prediction-feedback loop, no grad)
  GraphedRollout                the same step captured once in a hipGraph (static buffers; the
                                `torch.cat` window shift becomes an in-place roll)
  unrolled_train_iteration      ns_vorticity_unrolling.py:225-244 (look-ahead windows through the SOL
  / LookAheadCurriculum         wrapper, BPTT through n chained calls) and :216-223 (curriculum)
  central_diff / darcy_loss     exp_darcy.py:59-68, 209-234 (decode, rel-L2 + 0.1 x derivative loss,
  / darcy_train_step            clip, step) — the large-N single-call iteration
"""
from __future__ import annotations

import torch

from . import ops
from .model.Transolver_Structured_Mesh_2D import Model
from .utils.testloss import TestLoss


def build_model(cfg, state_dict=None, device="cuda", engine=None):
    """`engine`: GEMM engine of this model ("f32" | "split" | "bf16"; None = PA2D_GEMM / the split default)."""
    m = _build_model(cfg, state_dict, device)
    return m.set_engine(engine) if engine is not None else m


def model_engine(model):
    """The resolved GEMM engine id of a (possibly SOL-wrapped) Transolver."""
    inner = getattr(model, "transolver_model", model)
    return ops.resolve_engine(getattr(inner, "engine", None))


def _build_model(cfg, state_dict, device):
    m = Model(space_dim=cfg["space_dim"], n_layers=cfg["n_layers"], n_hidden=cfg["n_hidden"],
              dropout=cfg.get("dropout", 0.0), n_head=cfg["n_head"], Time_Input=cfg["Time_Input"],
              act=cfg.get("act", "gelu"), mlp_ratio=cfg["mlp_ratio"], fun_dim=cfg["fun_dim"],
              out_dim=cfg["out_dim"], slice_num=cfg["slice_num"], ref=cfg["ref"],
              unified_pos=cfg["unified_pos"], H=cfg["H"], W=cfg["W"])
    if state_dict is not None:
        m.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    return m.to(device)


def train_iteration(model, x, fx, yy, step=1, loss_fn=None, fold_time=False):
    """Forward part of one exp_ns mini-batch.  Returns (loss [graph attached], full_loss, pred).

    `fold_time`: the loop below is teacher-forced — the input window of call t is built from GROUND
    TRUTH frames only (exp_ns.py:205), so the T/step model calls do not depend on each other.  Folded,
    they run as ONE call on a batch of (T/step)*B windows: the same per-window outputs, the same summed
    loss and (up to fp32 summation order inside the weight-gradient reductions) the same gradients, with
    1/(T/step) of the launches, no per-call gradient accumulation and full tiles at the reference's small
    batch sizes.  Not applicable to the prediction-feedback rollout."""
    loss_fn = loss_fn or TestLoss(size_average=False)
    T = yy.shape[-1]
    bsz = x.shape[0]
    if fold_time:
        return _train_iteration_folded(model, x, fx, yy, step, loss_fn)
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


# widest per-row tensor any kernel addresses through a 32-bit buffer descriptor must stay below 4 GiB
FOLD_MAX_BYTES = 2 ** 32 - 4096


def _fold_group(model, bsz, npoints, ncalls):
    """How many teacher-forced calls fit in ONE folded model call: the widest activation row ([x_mid | fx_mid] =
    2C, the preprocess hidden 2C, or the MLP hidden r*C floats) times the folded row count must stay under the
    4 GiB extent of a buffer descriptor (libpa2d returns PA2D_ERR_UNSUPPORTED beyond it)."""
    inner = getattr(model, "transolver_model", model)
    C = inner.n_hidden
    r = max((blk.mlp.linear_pre[0].weight.shape[0] for blk in inner.blocks), default=C) / C
    width = int(max(2, r) * C)
    row_bytes = 4 * width
    eng = model_engine(model)
    if eng == ops.ENGINE_SPLIT:      # split engine: the 2C-wide conv gradient travels as 3 bf16 planes
        row_bytes = max(row_bytes, 6 * 2 * C)
    elif eng == ops.ENGINE_BF16S:    # bf16 storage: block activations are 2 bytes wide, the fp32 input embedding (2C) is not
        row_bytes = max(2 * width, 4 * 2 * C)
    rows = FOLD_MAX_BYTES // row_bytes
    return max(1, min(ncalls, rows // max(1, bsz * npoints)))


def _train_iteration_folded(model, x, fx, yy, step, loss_fn):
    T, bsz, F = yy.shape[-1], x.shape[0], fx.shape[-1]
    nt = len(range(0, T, step))
    seq = torch.cat((fx, yy), dim=-1)                                     # frames the windows slide over
    group = _fold_group(model, bsz, x.shape[1], nt)
    loss, outs = 0, []
    for t0 in range(0, nt, group):                                        # normally ONE group
        g = min(group, nt - t0)
        wins = torch.stack([seq[..., t * step:t * step + F] for t in range(t0, t0 + g)], 0)       # [g,B,N,F]
        im = model(x.repeat(g, 1, 1), fx=wins.reshape(g * bsz, *fx.shape[1:]))                    # [g*B,N,step]
        im = im.reshape(g, bsz, *im.shape[1:])
        for k in range(g):                    # per-call loss terms, so any reduction mode of loss_fn carries over
            t = t0 + k
            y = yy[..., t * step:(t + 1) * step]
            loss = loss + loss_fn(im[k].reshape(bsz, -1), y.reshape(bsz, -1))
        outs.append(im)
    im = torch.cat(outs, 0) if len(outs) > 1 else outs[0]
    pred = im.permute(1, 2, 0, 3).reshape(bsz, im.shape[2], -1)
    with torch.no_grad():
        full = loss_fn(pred.reshape(bsz, -1), yy.reshape(bsz, -1))
    return loss, full, pred


def train_step(model, optimizer, scheduler, x, fx, yy, step=1, max_grad_norm=None, grad_sync=None,
               set_to_none=False, loss_fn=None, fold_time=False):
    """One full exp_ns.py:191-218 iteration.  `grad_sync` (DDP): callable run between backward and
    the optimizer step (all-reduce SUM of the flat gradient bucket).  With `optim.FusedAdamW` pass
    `grad_sync=optimizer.sync` (same bucket) and put the clip threshold in the optimizer instead of
    `max_grad_norm`."""
    with ops.weights_frozen():       # no parameter moves between the first model call and the end of backward
        loss, full, _ = train_iteration(model, x, fx, yy, step, loss_fn, fold_time)
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
    with ops.weights_frozen():
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
        # the conv weight packs are made during the warm-up and only REFERENCED by the captured step; run()
        # refreshes them in place before replaying, so later weight updates (training between rollouts) are seen
        with ops.weights_frozen() as self.packs:
            with torch.cuda.stream(side), torch.no_grad():
                for _ in range(warmup):
                    self._step_eager()
            torch.cuda.current_stream().wait_stream(side)
            self.fx.copy_(fx)
            with torch.no_grad(), torch.cuda.graph(self.graph):
                self._step_eager()
        self.fx.copy_(fx)
        # the captured launches hold raw parameter pointers: anything that re-seats parameter storage afterwards
        # (an optimizer that flattens the parameters, .to(), load with assign=True) invalidates the graph
        self._param_ptrs = [p.data_ptr() for p in model.parameters()]

    def _step_eager(self):
        self.im = self.model(self.x, fx=self.fx)
        k = self.im.shape[-1]
        shifted = self.fx[..., self.step:].clone()
        self.fx[..., :-k].copy_(shifted[..., :self.fx.shape[-1] - k])
        self.fx[..., -k:].copy_(self.im)

    @torch.no_grad()
    def run(self, fx0, nsteps):
        if [p.data_ptr() for p in self.model.parameters()] != self._param_ptrs:
            raise RuntimeError("parameter storage moved since this rollout graph was captured (e.g. FusedAdamW / "
                               "FlatGradSync built afterwards, or .to()); build the optimizer first or re-capture")
        self.fx.copy_(fx0)
        self.packs.refresh()
        frames = []
        for _ in range(nsteps):
            self.graph.replay()
            frames.append(self.im.clone())
        return torch.cat(frames, -1)


class GraphedTrainStep:
    """exp_ns iteration with the launch-bound part captured in ONE hipGraph.

    At the reference's batch sizes (2-8) an iteration is ~14 000 kernel launches and the host cannot
    issue them as fast as the GPU retires them.  Captured once, replayed per iteration: zeroing the flat
    gradient bucket, the T/step teacher-forced model calls, the summed rel-L2 loss and the whole backward
    pass (static input buffers; every intermediate lives in the graph's private pool).  The gradient
    all-reduce, `optimizer.step()` and `scheduler.step()` stay eager because the OneCycle lr / beta1 are
    host scalars that change every iteration (2-3 launches).  Needs `optim.FusedAdamW` (gradients must be
    views of one persistent flat buffer).  Replays run the same kernels in the same order as the eager
    path, so results are bit-identical to `train_step`."""

    def __init__(self, model, optimizer, scheduler, x, fx, yy, step=1, loss_fn=None, warmup=3, fold_time=False):
        from .optim import FusedAdamW
        self.fold_time = fold_time
        if not isinstance(optimizer, FusedAdamW):
            raise TypeError("GraphedTrainStep needs optim.FusedAdamW (persistent flat gradient bucket)")
        self.model, self.opt, self.sched, self.step_size, self.loss_fn = model, optimizer, scheduler, step, loss_fn
        self.x, self.fx, self.yy = x.clone(), fx.clone(), yy.clone()
        snap = [p.detach().clone() for p in model.parameters()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):           # sets kernel attributes, warms the allocator
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        for p, q in zip(model.parameters(), snap):     # warm-up must not move the weights
            p.data.copy_(q)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.full = self._fwd_bwd()

    def _fwd_bwd(self):
        with ops.weights_frozen():      # inside the capture: each layer's weights are packed once per replay
            self.opt.zero_grad()
            loss, full, _ = train_iteration(self.model, self.x, self.fx, self.yy, self.step_size, self.loss_fn,
                                            self.fold_time)
            loss.backward()
        return loss.detach(), full

    def __call__(self, x, fx, yy):
        self.x.copy_(x)
        self.fx.copy_(fx)
        self.yy.copy_(yy)
        self.graph.replay()
        self.opt.sync()                 # all-reduce(SUM) of the flat bucket when world_size > 1
        self.opt.step()
        if self.sched is not None:
            self.sched.step()
        return self.loss, self.full


# ------------------------------------------------------------------------------ unrolled look-ahead training
def unrolled_train_iteration(sol_model, x, fx, yy, look_ahead, step=1, loss_fn=None):
    """One mini-batch of ns_vorticity_unrolling.py:225-244.  `sol_model.n` chained calls per window;
    the window then advances by `look_ahead` GROUND-TRUTH frames.  Returns the loss (graph attached)."""
    loss_fn = loss_fn or TestLoss(size_average=False)
    sol_model.n = look_ahead
    offset = step * look_ahead
    T = yy.shape[-1]
    bsz = x.shape[0]
    loss = 0
    for t in range(0, T - look_ahead + 1, look_ahead):
        y = yy[..., t + offset - step:t + offset]
        im = sol_model(x, fx)
        loss = loss + loss_fn(im.reshape(bsz, -1), y.reshape(bsz, -1))
        fx = torch.cat((fx[..., look_ahead:], yy[..., t:t + look_ahead]), dim=-1)
    return loss


class LookAheadCurriculum:
    """look_ahead doubles (capped) whenever `ep % thresh == 0 and ep >= thresh`, after which the
    threshold halves (ns_vorticity_unrolling.py:205-223)."""

    def __init__(self, epochs, look_ahead=1, max_look_ahead=10):
        self.look_ahead, self.max_look_ahead, self.thresh = look_ahead, max_look_ahead, epochs / 2

    def update(self, ep):
        if ep % self.thresh == 0 and ep >= self.thresh and self.look_ahead <= self.max_look_ahead:
            self.look_ahead = min(self.look_ahead * 2, self.max_look_ahead)
            self.thresh /= 2
        return self.look_ahead


# ------------------------------------------------------------------------------ Darcy iteration
def central_diff(x, h, resolution):
    """x: [B, res*res, C]; zero-padded central differences along image x / y (exp_darcy.py:59-68)."""
    B, N, C = x.shape
    img = torch.nn.functional.pad(x.reshape(B, resolution, resolution, C), (0, 0, 1, 1, 1, 1))
    gx = (img[:, 1:-1, 2:, :] - img[:, 1:-1, :-2, :]) / (2 * h)
    gy = (img[:, 2:, 1:-1, :] - img[:, :-2, 1:-1, :]) / (2 * h)
    return gx, gy


def darcy_loss(out, y, y_normalizer, dx, s, loss_fn=None):
    """out, y: [B, N] normalised prediction / target.  Returns (loss, l2loss, deriv_loss) with
    loss = l2 + 0.1 * (rel-L2 of d/dx + rel-L2 of d/dy), prediction border zeroed first."""
    loss_fn = loss_fn or TestLoss(size_average=False)
    out = y_normalizer.decode(out)
    y = y_normalizer.decode(y)
    l2 = loss_fn(out, y)
    B = out.shape[0]
    img = out.reshape(B, s, s)
    inner = torch.zeros_like(img)
    inner[:, 1:-1, 1:-1] = img[:, 1:-1, 1:-1]
    gtx, gty = central_diff(y.unsqueeze(-1), dx, s)
    px, py = central_diff(inner.reshape(B, s * s, 1), dx, s)
    deriv = loss_fn(px, gtx) + loss_fn(py, gty)
    return 0.1 * deriv + l2, l2, deriv


def darcy_train_step(model, optimizer, scheduler, x, fx, y, y_normalizer, dx, s, max_grad_norm=None,
                     grad_sync=None):
    """exp_darcy.py:209-234: ONE model call per iteration (fun_dim = 1)."""
    optimizer.zero_grad(set_to_none=False)
    out = model(x, fx=fx.unsqueeze(-1)).squeeze(-1)
    loss, l2, deriv = darcy_loss(out, y, y_normalizer, dx, s)
    loss.backward()
    if grad_sync is not None:
        grad_sync()
    if max_grad_norm is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_grad_norm)
    optimizer.step()
    if scheduler is not None:
        scheduler.step()
    return loss.detach(), l2.detach(), deriv.detach()
