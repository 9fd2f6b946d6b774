"""Forward/backward compositions of the libpa2d stages and their autograd wrappers.

One autograd node per stage group (LayerNorm, Physics-Attention incl. to_out + residual, MLP incl.
residual, head) instead of the ~120 nodes per layer the reference builds; the slice
weights [B,h,N,M] are never saved — backward recomputes them from x_mid inside the kernels.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops


# ------------------------------------------------------------------------------ Physics-Attention
def attn_forward(xn, P, res, H, W, heads):
    """xn [B,N,C] (already layer-normed).  P: dict of parameter tensors.  Returns (out, saved).
    H is None -> irregular-mesh variant (Physics_Attention.py:6-57): Linear projections, no
    temperature clamp; otherwise the structured-mesh variant (3x3 conv projections, clamp)."""
    B, N, C = xn.shape
    D = C // heads
    M = P["ws"].shape[0]
    temp = P["temperature"].reshape(heads).contiguous()
    structured = H is not None
    if structured:
        xf = ops.conv3x3x2_fwd(xn, P["wx"], P["bx"], P["wf"], P["bf"], H, W)                   # [B,N,2C]
    else:   # both Linear(C, C) projections as ONE GEMM with the weights stacked along the output dim
        wcat, bcat = torch.cat((P["wx"], P["wf"]), 0), torch.cat((P["bx"], P["bf"]), 0)
        xf = ops.linear_fwd(xn.view(B * N, C), wcat, bcat)[0].view(B, N, 2 * C)
    spart, npart = ops.slice_scatter(xf, 2 * C, 0, xf, 2 * C, C, P["ws"], P["bs"], temp, B, N, heads, D, M,
                                     clamp=structured)
    s, nrm, o = ops.token_attn_fwd(spart, npart, P["wq"], P["wk"], P["wv"])
    y = ops.deslice_fwd(xf, 2 * C, 0, o, P["ws"], P["bs"], temp, B, N, heads, D, M, clamp=structured)  # [B,N,C]
    out, _ = ops.linear_fwd(y.view(B * N, C), P["wo"], P["bo"],
                            res=None if res is None else res.reshape(B * N, C))
    return out.view(B, N, C), (xn, xf, s, nrm, o, y, temp)


def attn_backward(saved, P, dout, H, W, heads, need_dx=True):
    xn, xf, s, nrm, o, y, temp = saved
    B, N, C = xn.shape
    D = C // heads
    M = P["ws"].shape[0]
    d2 = dout.reshape(B * N, C)
    dy = ops.linear_bwd_data(d2, P["wo"])                                                     # [B*N,C]
    dwo, dbo = ops.linear_bwd_weight(d2, y.view(B * N, C))
    structured = H is not None
    dopart, _ = ops.slice_scatter(xf, 2 * C, 0, dy, C, 0, P["ws"], P["bs"], temp, B, N, heads, D, M,
                                  want_norm=False, clamp=structured)
    ds, dn, dwq, dwk, dwv = ops.token_attn_bwd(s, nrm, P["wq"], P["wk"], P["wv"], dopart)
    dxf, dws, dbs, dtemp = ops.slice_bwd_points(xf, dy, P["ws"], P["bs"], temp, o, ds, dn, B, N, heads, D, M,
                                                clamp=structured)
    if structured:
        dxn, dwx, dbx, dwf, dbf = ops.conv3x3x2_bwd(dxf, xn, P["wx"], P["wf"], H, W, need_dx=need_dx)
    else:
        dxf2, xn2 = dxf.view(B * N, 2 * C), xn.view(B * N, C)
        dwcat, dbcat = ops.linear_bwd_weight(dxf2, xn2)
        dwx, dwf, dbx, dbf = dwcat[:C].contiguous(), dwcat[C:].contiguous(), dbcat[:C].contiguous(), dbcat[C:].contiguous()
        dxn = ops.linear_bwd_data(dxf2, torch.cat((P["wx"], P["wf"]), 0)).view(B, N, C) if need_dx else None
    grads = dict(temperature=dtemp.view(1, heads, 1, 1), wx=dwx, bx=dbx, wf=dwf, bf=dbf, ws=dws, bs=dbs,
                 wq=dwq, wk=dwk, wv=dwv, wo=dwo, bo=dbo)
    return dxn, grads


ATTN_KEYS = ("temperature", "wx", "bx", "wf", "bf", "ws", "bs", "wq", "wk", "wv", "wo", "bo")


class PhysicsAttentionFn(Function):
    """out = to_out(deslice(attn(slice(conv(xn))))) (+ res)."""

    @staticmethod
    def forward(ctx, xn, res, H, W, heads, *params):
        P = dict(zip(ATTN_KEYS, (p.detach().contiguous() for p in params)))
        out, saved = attn_forward(xn.detach().contiguous(), P, None if res is None else res.detach().contiguous(),
                                  H, W, heads)
        ctx.P, ctx.saved, ctx.geom = P, saved, (H, W, heads)
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        H, W, heads = ctx.geom
        dout = dout.contiguous()
        dxn, g = attn_backward(ctx.saved, ctx.P, dout, H, W, heads, need_dx=ctx.needs_input_grad[0])
        return (dxn, dout if ctx.has_res else None, None, None, None) + tuple(g[k] for k in ATTN_KEYS)


# ------------------------------------------------------------------------------ MLP (Linear-act-Linear)
def mlp_forward(x2d, w1, b1, w2, b2, act, res2d, need_bwd=True):
    """need_bwd=False (inference): the pre-activation is not written (one [rows, r*C] store less)."""
    hact, hpre = ops.linear_fwd(x2d, w1, b1, act=act, want_pre=need_bwd)
    out, _ = ops.linear_fwd(hact, w2, b2, res=res2d)
    return out, (x2d, hpre, hact)


def mlp_backward(saved, w1, w2, act, dout2d, need_dx=True):
    x2d, hpre, hact = saved
    dhpre = ops.linear_bwd_data(dout2d, w2, pre=hpre, act=act)
    dw2, db2 = ops.linear_bwd_weight(dout2d, hact)
    dw1, db1 = ops.linear_bwd_weight(dhpre, x2d)
    dx = ops.linear_bwd_data(dhpre, w1) if need_dx else None
    return dx, dw1, db1, dw2, db2


class MLPFn(Function):
    """linear_post(act(linear_pre(x))) (+ res) for the n_layers=0 MLP of the reference."""

    @staticmethod
    def forward(ctx, x, res, act, w1, b1, w2, b2):
        shp = x.shape
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        w1, b1, w2, b2 = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        res2d = None if res is None else res.detach().reshape(-1, w2.shape[0]).contiguous()
        out, saved = mlp_forward(x2d, w1, b1, w2, b2, act, res2d, need_bwd=any(ctx.needs_input_grad))
        ctx.saved, ctx.w, ctx.act, ctx.shp, ctx.has_res = saved, (w1, w2), act, shp, res is not None
        return out.view(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dout):
        w1, w2 = ctx.w
        d2 = dout.reshape(-1, w2.shape[0]).contiguous()
        dx, dw1, db1, dw2, db2 = mlp_backward(ctx.saved, w1, w2, ctx.act, d2, need_dx=ctx.needs_input_grad[0])
        return (None if dx is None else dx.view(ctx.shp), dout if ctx.has_res else None, None, dw1, db1, dw2, db2)


# ------------------------------------------------------------------------------ LayerNorm / head
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        shp = x.shape
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2d, gamma, beta)
        ctx.saved, ctx.shp = (x2d, mean, rstd, gamma), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd, gamma = ctx.saved
        dx, dg, db = ops.layernorm_bwd(dy.reshape(x2d.shape).contiguous(), x2d, mean, rstd, gamma)
        return dx.view(ctx.shp), dg, db


class HeadFn(Function):
    """mlp2: Linear(C, out_dim <= 8)."""

    @staticmethod
    def forward(ctx, xn, w, b):
        shp = xn.shape
        x2d = xn.detach().reshape(-1, shp[-1]).contiguous()
        w, b = w.detach().contiguous(), b.detach().contiguous()
        ctx.saved, ctx.shp = (x2d, w), shp
        return ops.head_fwd(x2d, w, b).view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved
        dxn, dw, db = ops.head_bwd(dy.reshape(-1, w.shape[0]).contiguous(), x2d, w)
        return dxn.view(ctx.shp), dw, db


class AttnBranchFn(Function):
    """fx + Attn(LayerNorm(fx)) as ONE autograd node (Transolver_block.forward, …_2D.py:70): the residual
    gradient is folded into the LayerNorm backward kernel (`dres`), so no separate elementwise add runs."""

    @staticmethod
    def forward(ctx, fx, ln_w, ln_b, H, W, heads, *params):
        shp = fx.shape
        fx2d = fx.detach().reshape(-1, shp[-1]).contiguous()
        ln_w, ln_b = ln_w.detach().contiguous(), ln_b.detach().contiguous()
        xn, mean, rstd = ops.layernorm_fwd(fx2d, ln_w, ln_b)
        P = dict(zip(ATTN_KEYS, (p.detach().contiguous() for p in params)))
        out, saved = attn_forward(xn.view(shp), P, fx2d.view(shp), H, W, heads)
        ctx.P, ctx.saved, ctx.geom, ctx.ln = P, saved, (H, W, heads), (fx2d, mean, rstd, ln_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        H, W, heads = ctx.geom
        fx2d, mean, rstd, ln_w = ctx.ln
        dout = dout.contiguous()
        dxn, g = attn_backward(ctx.saved, ctx.P, dout, H, W, heads, need_dx=True)
        dfx, dg, db = ops.layernorm_bwd(dxn.reshape(fx2d.shape), fx2d, mean, rstd, ln_w, dres=dout.reshape(fx2d.shape))
        return (dfx.view(dout.shape), dg, db, None, None, None) + tuple(g[k] for k in ATTN_KEYS)


class MLPBranchFn(Function):
    """fx + MLP(LayerNorm(fx)) as one autograd node (…_2D.py:71), residual gradient folded into LN backward."""

    @staticmethod
    def forward(ctx, fx, ln_w, ln_b, act, w1, b1, w2, b2):
        shp = fx.shape
        fx2d = fx.detach().reshape(-1, shp[-1]).contiguous()
        ln_w, ln_b, w1, b1, w2, b2 = (t.detach().contiguous() for t in (ln_w, ln_b, w1, b1, w2, b2))
        xn, mean, rstd = ops.layernorm_fwd(fx2d, ln_w, ln_b)
        out, saved = mlp_forward(xn, w1, b1, w2, b2, act, fx2d, need_bwd=any(ctx.needs_input_grad))
        ctx.saved, ctx.w, ctx.act, ctx.ln = saved, (w1, w2), act, (fx2d, mean, rstd, ln_w)
        return out.view(shp)

    @staticmethod
    def backward(ctx, dout):
        w1, w2 = ctx.w
        fx2d, mean, rstd, ln_w = ctx.ln
        d2 = dout.reshape(fx2d.shape).contiguous()
        dxn, dw1, db1, dw2, db2 = mlp_backward(ctx.saved, w1, w2, ctx.act, d2, need_dx=True)
        dfx, dg, db = ops.layernorm_bwd(dxn, fx2d, mean, rstd, ln_w, dres=d2)
        return dfx.view(dout.shape), dg, db, None, dw1, db1, dw2, db2


def attn_branch(fx, ln_w, ln_b, H, W, heads, params):
    return AttnBranchFn.apply(fx, ln_w, ln_b, H, W, heads, *params)


def mlp_branch(fx, ln_w, ln_b, act, w1, b1, w2, b2):
    return MLPBranchFn.apply(fx, ln_w, ln_b, act, w1, b1, w2, b2)


class LinearFn(Function):
    """y = act(x . w^T + b): generic dense layer (MLP hidden layers when n_layers > 0, wide heads)."""

    @staticmethod
    def forward(ctx, x, act, w, b):
        shp = x.shape
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        w = w.detach().contiguous()
        b = None if b is None else b.detach().contiguous()
        y, pre = ops.linear_fwd(x2d, w, b, act=act, want_pre=act is not None)
        ctx.saved, ctx.act, ctx.shp, ctx.has_b = (x2d, w, pre), act, shp, b is not None
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, w, pre = ctx.saved
        d2 = dy.reshape(-1, w.shape[0]).contiguous()
        if ctx.act is not None:
            d2 = ops.act_bwd(d2, pre, ctx.act)
        dw, db = ops.linear_bwd_weight(d2, x2d, want_bias=ctx.has_b)
        dx = ops.linear_bwd_data(d2, w).view(ctx.shp) if ctx.needs_input_grad[0] else None
        return dx, None, dw, db


def linear(x, w, b, act):
    return LinearFn.apply(x, act, w, b)


def layer_norm(x, gamma, beta):
    return LayerNormFn.apply(x, gamma, beta)


def physics_attention(xn, res, H, W, heads, params):
    return PhysicsAttentionFn.apply(xn, res, H, W, heads, *params)


def mlp(x, res, act, w1, b1, w2, b2):
    return MLPFn.apply(x, res, act, w1, b1, w2, b2)


def head(xn, w, b):
    return HeadFn.apply(xn, w, b)


class RelL2Fn(Function):
    """Per-sample ||pred-y||_2 / ||y||_2 (utils/testloss.py:31-42) in one kernel; gradient w.r.t. pred."""

    @staticmethod
    def forward(ctx, pred, y):
        B = pred.shape[0]
        p2, y2 = pred.detach().reshape(B, -1).contiguous(), y.detach().reshape(B, -1).contiguous()
        dn, yn, ratio = ops.rel_l2_fwd(p2, y2)
        ctx.saved, ctx.shp = (p2, y2, dn, yn), pred.shape
        return ratio

    @staticmethod
    def backward(ctx, gratio):
        p2, y2, dn, yn = ctx.saved
        # d ratio_b / d pred = (pred - y) / (dn_b * yn_b); fold the incoming per-sample gradient into yn
        g = gratio.contiguous()
        one = torch.ones(1, dtype=torch.float32, device=p2.device)
        dpred = ops.rel_l2_bwd(p2, y2, dn, yn / g, one)
        return dpred.view(ctx.shp), None


def rel_l2(pred, y):
    return RelL2Fn.apply(pred, y)
