"""ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement (plain PyTorch, fp32 or fp64, op-for-op unfused) of the reference's Transolver
structured-mesh-2D hot path, written from the math contract in SURVEY.md Appendix A.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
the shipped package never does (it fails loudly without its HIP library instead).

Parity status: PINNED for the model, the attention block, the SOL wrapper, the loss, the exp_ns iteration and the
rollout: `oracle/make_golden.py` imports the reference itself in the build container and (a) asserts this restatement
reproduces it (forward and every parameter gradient) and (b) writes the golden vectors G1-G7 under `tests/golden/`
that `tests/test_oracle_golden.py` re-checks everywhere.  The reference has no tests / fixtures of its own.
PARITY UNPINNED: `unrolled_iteration_loss`, `darcy_loss`, `central_diff` — the reference DRIVERS that contain these
loops execute argparse and a hard-coded data path at import and cannot be imported, so they are restated from the
source text only (their building blocks are pinned).

Reference lines each function follows (relative to the reference repo root):
  unified_pos            model/Transolver_Structured_Mesh_2D.py:183-200
  layer_norm             nn.LayerNorm use at model/Transolver_Structured_Mesh_2D.py:58,62,65
  conv3x3                nn.Conv2d(C, C, 3, 1, 1) at model/Physics_Attention.py:74-75,94,96
  slice_tokens           model/Physics_Attention.py:94-102
  token_attention        model/Physics_Attention.py:105-111
  deslice                model/Physics_Attention.py:116-117
  physics_attention      model/Physics_Attention.py:88-119
  mlp                    model/Transolver_Structured_Mesh_2D.py:30-38
  block                  model/Transolver_Structured_Mesh_2D.py:69-75
  model_forward          model/Transolver_Structured_Mesh_2D.py:202-220
  sol_forward            model/SOL_Transolver_Structured_Mesh_2D.py:47-52
  rel_l2                 utils/testloss.py:31-42
  train_iteration        exp_ns.py:191-218
  rollout                exp_ns.py:225-241, ns_vorticity_unrolling.py:264-286
  timestep_embedding     model/Embedding.py:67-85
  physics_attention_irregular, model_forward_irregular
                         model/Physics_Attention.py:6-57, model/Transolver_Irregular_Mesh.py:137-158
  unrolled_iteration_loss  ns_vorticity_unrolling.py:225-244
  central_diff, darcy_loss exp_darcy.py:59-68, 216-227
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F

SLICE_EPS = 1e-5   # Physics_Attention.py:102
LN_EPS = 1e-5      # nn.LayerNorm default
TAU_MIN, TAU_MAX = 0.1, 5.0   # Physics_Attention.py:99


def to_torch(sd, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(np.asarray(v) if not torch.is_tensor(v) else v).detach().to(dtype).clone()
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def infer_config(sd):
    """Recover the architecture hyper-parameters that are visible in state_dict shapes."""
    C = sd["placeholder"].shape[0]
    L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    h = sd["blocks.0.Attn.temperature"].shape[1]
    M = sd["blocks.0.Attn.in_project_slice.weight"].shape[0]
    r = sd["blocks.0.mlp.linear_pre.0.weight"].shape[0] // C
    out_dim = sd[f"blocks.{L - 1}.mlp2.weight"].shape[0]
    return dict(n_hidden=C, n_layers=L, n_head=h, slice_num=M, mlp_ratio=r, out_dim=out_dim,
                in_features=sd["preprocess.linear_pre.0.weight"].shape[1],
                Time_Input=("time_fc.0.weight" in sd))


# ----------------------------------------------------------------------------- building blocks
def unified_pos(H, W, ref, dtype=torch.float32):
    """[1, H*W, ref*ref]: distance of every grid point to a ref x ref lattice.  linspace is taken
    in float64 then cast to float32 before the arithmetic."""
    gy = torch.tensor(np.linspace(0, 1, H), dtype=torch.float32)
    gx = torch.tensor(np.linspace(0, 1, W), dtype=torch.float32)
    ry = torch.tensor(np.linspace(0, 1, ref), dtype=torch.float32)
    rx = torch.tensor(np.linspace(0, 1, ref), dtype=torch.float32)
    d0 = gy[:, None, None, None] - ry[None, None, :, None]        # H,1,ref,1
    d1 = gx[None, :, None, None] - rx[None, None, None, :]        # 1,W,1,ref
    pos = torch.sqrt(d0 ** 2 + d1 ** 2).reshape(1, H * W, ref * ref)
    return pos.to(dtype)


def layer_norm(x, g, b):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def conv3x3(x_bnc, w, b, H, W):
    """Zero-padded 3x3 cross-correlation on the [B, H*W, C] (== NHWC) view of the field."""
    B, N, C = x_bnc.shape
    img = x_bnc.reshape(B, H, W, C).permute(0, 3, 1, 2)
    out = F.conv2d(img, w, b, stride=1, padding=1)
    return out.permute(0, 2, 3, 1).reshape(B, N, -1)


def split_heads(t, h):
    B, N, C = t.shape
    return t.reshape(B, N, h, C // h).permute(0, 2, 1, 3)      # B,h,N,D


def slice_weights(xm_h, ws, bs, temperature):
    tau = temperature.reshape(1, -1, 1, 1).clamp(TAU_MIN, TAU_MAX)
    logits = (xm_h @ ws.t() + bs) / tau
    return torch.softmax(logits, dim=-1)                        # B,h,N,M


def slice_tokens(xm, fm, ws, bs, temperature, h):
    """-> (W [B,h,N,M], norm [B,h,M], raw sums S [B,h,M,D], tokens T [B,h,M,D])"""
    w = slice_weights(split_heads(xm, h), ws, bs, temperature)
    norm = w.sum(2)
    s = w.transpose(-1, -2) @ split_heads(fm, h)
    return w, norm, s, s / (norm + SLICE_EPS)[..., None]


def token_attention(tok, wq, wk, wv):
    D = tok.shape[-1]
    q, k, v = tok @ wq.t(), tok @ wk.t(), tok @ wv.t()
    a = torch.softmax((q @ k.transpose(-1, -2)) * D ** -0.5, dim=-1)
    return a @ v


def deslice(w, o):
    y = w @ o                                                   # B,h,N,D
    B, h, N, D = y.shape
    return y.permute(0, 2, 1, 3).reshape(B, N, h * D)


def physics_attention(xn, sd, pre, H, W, h, want=None):
    """to_out(deslice(attn(slice(conv(xn))))).  `want`: optional dict filled with intermediates."""
    xm = conv3x3(xn, sd[pre + "in_project_x.weight"], sd[pre + "in_project_x.bias"], H, W)
    fm = conv3x3(xn, sd[pre + "in_project_fx.weight"], sd[pre + "in_project_fx.bias"], H, W)
    w, norm, s, tok = slice_tokens(xm, fm, sd[pre + "in_project_slice.weight"],
                                   sd[pre + "in_project_slice.bias"], sd[pre + "temperature"], h)
    o = token_attention(tok, sd[pre + "to_q.weight"], sd[pre + "to_k.weight"], sd[pre + "to_v.weight"])
    y = deslice(w, o)
    out = y @ sd[pre + "to_out.0.weight"].t() + sd[pre + "to_out.0.bias"]
    if want is not None:
        want.update(xm=xm, fm=fm, w=w, norm=norm, s=s, tok=tok, o=o, y=y, out=out)
    return out


def physics_attention_irregular(xn, sd, pre, h):
    """model/Physics_Attention.py:6-57: Linear projections, temperature NOT clamped."""
    xm = xn @ sd[pre + "in_project_x.weight"].t() + sd[pre + "in_project_x.bias"]
    fm = xn @ sd[pre + "in_project_fx.weight"].t() + sd[pre + "in_project_fx.bias"]
    logits = (split_heads(xm, h) @ sd[pre + "in_project_slice.weight"].t() + sd[pre + "in_project_slice.bias"]) \
        / sd[pre + "temperature"].reshape(1, -1, 1, 1)
    w = torch.softmax(logits, dim=-1)
    norm = w.sum(2)
    tok = (w.transpose(-1, -2) @ split_heads(fm, h)) / (norm + SLICE_EPS)[..., None]
    o = token_attention(tok, sd[pre + "to_q.weight"], sd[pre + "to_k.weight"], sd[pre + "to_v.weight"])
    return deslice(w, o) @ sd[pre + "to_out.0.weight"].t() + sd[pre + "to_out.0.bias"]


def model_forward_irregular(sd, x, fx, cfg, T=None):
    """model/Transolver_Irregular_Mesh.py:137-158 (placeholder ALWAYS added; unified_pos from x)."""
    h, act = cfg["n_head"], cfg.get("act", "gelu")
    if cfg["unified_pos"]:
        ref = cfg["ref"]
        r = torch.tensor(np.linspace(0, 1, ref), dtype=torch.float32).to(x.dtype).to(x.device)
        grid = torch.stack((r.reshape(ref, 1).expand(ref, ref), r.reshape(1, ref).expand(ref, ref)), -1).reshape(1, ref * ref, 2)
        x = torch.sqrt(((x[:, :, None, :] - grid[:, None, :, :]) ** 2).sum(-1))
    z = mlp(torch.cat((x, fx), -1) if fx is not None else x, sd, "preprocess.", act) + sd["placeholder"][None, None, :]
    if T is not None:
        C = sd["placeholder"].shape[0]
        e = timestep_embedding(T, C).to(z.dtype)
        e = F.silu(e @ sd["time_fc.0.weight"].t() + sd["time_fc.0.bias"])
        z = z + (e @ sd["time_fc.2.weight"].t() + sd["time_fc.2.bias"])
    L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(L):
        p = f"blocks.{i}."
        z = physics_attention_irregular(layer_norm(z, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), sd, p + "Attn.", h) + z
        z = mlp(layer_norm(z, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), sd, p + "mlp.", act) + z
        if i == L - 1:
            z = layer_norm(z, sd[p + "ln_3.weight"], sd[p + "ln_3.bias"]) @ sd[p + "mlp2.weight"].t() + sd[p + "mlp2.bias"]
    return z


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


_ACTS = {
    "gelu": gelu_erf, "tanh": torch.tanh, "sigmoid": torch.sigmoid, "relu": torch.relu,
    "softplus": F.softplus, "ELU": F.elu, "silu": F.silu,
}


def mlp(x, sd, pre, act="gelu"):
    hdn = _ACTS[act](x @ sd[pre + "linear_pre.0.weight"].t() + sd[pre + "linear_pre.0.bias"])
    return hdn @ sd[pre + "linear_post.weight"].t() + sd[pre + "linear_post.bias"]


def block(fx, sd, i, H, W, h, last, act="gelu"):
    p = f"blocks.{i}."
    fx = physics_attention(layer_norm(fx, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]), sd,
                           p + "Attn.", H, W, h) + fx
    fx = mlp(layer_norm(fx, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"]), sd, p + "mlp.", act) + fx
    if last:
        z = layer_norm(fx, sd[p + "ln_3.weight"], sd[p + "ln_3.bias"])
        return z @ sd[p + "mlp2.weight"].t() + sd[p + "mlp2.bias"]
    return fx


def timestep_embedding(t, dim, max_period=10000):
    """t: [B,1] (exp_plas.py:186 passes `tim[:, t:t+1].reshape(bsz, 1)`) -> [B,1,dim], always
    computed in float32 like the reference."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    args = t.reshape(-1, 1, 1).float() * freqs[None, None, :]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[..., :1])], dim=-1)
    return emb


def model_forward(sd, x, fx, cfg, T=None):
    """cfg needs: H, W, ref, unified_pos, n_head (rest is read off the state_dict shapes)."""
    H, W, h = cfg["H"], cfg["W"], cfg["n_head"]
    act = cfg.get("act", "gelu")
    dtype = sd["placeholder"].dtype
    B = x.shape[0]
    if cfg["unified_pos"]:
        x = unified_pos(H, W, cfg["ref"], dtype).expand(B, -1, -1)
    if fx is not None:
        z = mlp(torch.cat((x, fx), -1), sd, "preprocess.", act)
    else:
        z = mlp(x, sd, "preprocess.", act) + sd["placeholder"][None, None, :]
    if T is not None:
        C = sd["placeholder"].shape[0]
        e = timestep_embedding(T, C).to(dtype)
        e = F.silu(e @ sd["time_fc.0.weight"].t() + sd["time_fc.0.bias"])
        z = z + (e @ sd["time_fc.2.weight"].t() + sd["time_fc.2.bias"])
    L = 1 + max(int(k.split(".")[1]) for k in sd if k.startswith("blocks."))
    for i in range(L):
        z = block(z, sd, i, H, W, h, last=(i == L - 1), act=act)
    return z


def sol_forward(sd, x, fx, cfg, n, step=1):
    u = None
    for _ in range(n):
        u = model_forward(sd, x, fx, cfg)
        fx = torch.cat((fx[..., step:], u), dim=-1)
    return u


def rel_l2(pred, y, size_average=False, reduction=True):
    B = pred.shape[0]
    d = torch.linalg.vector_norm(pred.reshape(B, -1) - y.reshape(B, -1), 2, dim=1)
    n = torch.linalg.vector_norm(y.reshape(B, -1), 2, dim=1)
    if not reduction:
        return d / n
    return (d / n).mean() if size_average else (d / n).sum()


def train_iteration(sd, x, fx, yy, cfg, step=1):
    """One exp_ns.py mini-batch: T/step teacher-forced calls, summed rel-L2, one backward.
    `sd` tensors must have requires_grad=True.  Returns (loss, full_loss, preds, grads)."""
    T = yy.shape[-1]
    B = x.shape[0]
    loss = 0.0
    preds = []
    for t in range(0, T, step):
        y = yy[..., t:t + step]
        im = model_forward(sd, x, fx, cfg)
        loss = loss + rel_l2(im.reshape(B, -1), y.reshape(B, -1))
        preds.append(im)
        fx = torch.cat((fx[..., step:], y), dim=-1)
    pred = torch.cat(preds, -1)
    full = rel_l2(pred.reshape(B, -1), yy.reshape(B, -1))
    keys = [k for k in sd if sd[k].requires_grad]
    gs = torch.autograd.grad(loss, [sd[k] for k in keys], allow_unused=True)
    grads = {k: g for k, g in zip(keys, gs)}
    return loss.detach(), full.detach(), pred.detach(), grads


@torch.no_grad()
def rollout(sd, x, fx, cfg, nsteps, step=1):
    """Prediction-feedback loop; returns [B,N,nsteps*out_dim]."""
    frames = []
    for _ in range(nsteps):
        im = model_forward(sd, x, fx, cfg)
        frames.append(im)
        fx = torch.cat((fx[..., step:], im), dim=-1)
    return torch.cat(frames, -1)


def unrolled_iteration_loss(sd, x, fx, yy, cfg, look_ahead, step=1):
    """ns_vorticity_unrolling.py:225-244: windows of `look_ahead` chained calls, loss on the last
    prediction of each window, window advanced with ground truth."""
    T = yy.shape[-1]
    B = x.shape[0]
    offset = step * look_ahead
    loss = 0.0
    for t in range(0, T - look_ahead + 1, look_ahead):
        y = yy[..., t + offset - step:t + offset]
        im = sol_forward(sd, x, fx, cfg, look_ahead, step)
        loss = loss + rel_l2(im.reshape(B, -1), y.reshape(B, -1))
        fx = torch.cat((fx[..., look_ahead:], yy[..., t:t + look_ahead]), dim=-1)
    return loss


def central_diff(x, h, res):
    """exp_darcy.py:59-68"""
    B, N, C = x.shape
    img = F.pad(x.reshape(B, res, res, C), (0, 0, 1, 1, 1, 1))
    return ((img[:, 1:-1, 2:] - img[:, 1:-1, :-2]) / (2 * h), (img[:, 2:, 1:-1] - img[:, :-2, 1:-1]) / (2 * h))


def darcy_loss(out, y, mean, std, dx, s):
    """exp_darcy.py:216-227 with UnitTransformer.decode (utils/normalizer.py:52-53)."""
    out, y = out * std + mean, y * std + mean
    l2 = rel_l2(out, y)
    B = out.shape[0]
    mask = torch.zeros(s, s, dtype=out.dtype, device=out.device)
    mask[1:-1, 1:-1] = 1
    inner = (out.reshape(B, s, s) * mask).reshape(B, s * s, 1)
    gx, gy = central_diff(y.unsqueeze(-1), dx, s)
    px, py = central_diff(inner, dx, s)
    deriv = rel_l2(px, gx) + rel_l2(py, gy)
    return 0.1 * deriv + l2, l2, deriv


# ----------------------------------------------------------------------------- A.2 backward (hand-derived)
def slice_core_backward(xm, fm, dy, ws, bs, temperature, wq, wk, wv, h):
    """Explicit backward of slice -> token attention -> de-slice, used to
    test the HIP backward kernels stage by stage (also checked against autograd in the CPU tests).

    xm, fm, dy: [B,N,C] (dy = gradient w.r.t. the de-sliced [B,N,C] tensor `y`).
    Returns dict(dxm, dfm, dws, dbs, dtemperature, dwq, dwk, dwv, do, ds, dn)."""
    B, N, C = xm.shape
    D = C // h
    xh, fh, dyh = split_heads(xm, h), split_heads(fm, h), split_heads(dy, h)
    traw = temperature.reshape(1, h, 1, 1)
    tau = traw.clamp(TAU_MIN, TAU_MAX)
    logit = (xh @ ws.t() + bs) / tau
    w = torch.softmax(logit, -1)
    n = w.sum(2)                                                # B,h,M
    s = w.transpose(-1, -2) @ fh                                # B,h,M,D
    t = s / (n + SLICE_EPS)[..., None]
    q, k, v = t @ wq.t(), t @ wk.t(), t @ wv.t()
    sc = D ** -0.5
    a = torch.softmax(q @ k.transpose(-1, -2) * sc, -1)
    o = a @ v
    # de-slice
    do = w.transpose(-1, -2) @ dyh                              # B,h,M,D
    dw1 = dyh @ o.transpose(-1, -2)                             # B,h,N,M
    # token attention
    da = do @ v.transpose(-1, -2)
    dv = a.transpose(-1, -2) @ do
    dp = a * (da - (da * a).sum(-1, keepdim=True))
    dq = dp @ k * sc
    dk = dp.transpose(-1, -2) @ q * sc
    dt = dq @ wq + dk @ wk + dv @ wv
    dwq = (dq.transpose(-1, -2) @ t).sum((0, 1))
    dwk = (dk.transpose(-1, -2) @ t).sum((0, 1))
    dwv = (dv.transpose(-1, -2) @ t).sum((0, 1))
    # token normalisation
    ds = dt / (n + SLICE_EPS)[..., None]
    dn = -(dt * s).sum(-1) / (n + SLICE_EPS) ** 2               # B,h,M
    # slice
    dfm = w @ ds                                                # B,h,N,D
    dw = dw1 + fh @ ds.transpose(-1, -2) + dn[:, :, None, :]
    dl = w * (dw - (dw * w).sum(-1, keepdim=True))
    dxm = (dl @ ws) / tau
    dws = ((dl / tau).transpose(-1, -2) @ xh).sum((0, 1))
    dbs = (dl / tau).sum((0, 1, 2))
    dtau = -(dl * logit).sum((0, 2, 3)) / tau.reshape(h)
    inside = ((traw.reshape(h) >= TAU_MIN) & (traw.reshape(h) <= TAU_MAX)).to(dtau.dtype)
    merge = lambda z: z.permute(0, 2, 1, 3).reshape(B, N, C)
    return dict(dxm=merge(dxm), dfm=merge(dfm), dws=dws, dbs=dbs,
                dtemperature=(dtau * inside).reshape(1, h, 1, 1), dwq=dwq, dwk=dwk, dwv=dwv,
                do=do, ds=ds, dn=dn)
