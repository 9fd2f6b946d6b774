"""Forward/backward compositions of the libpa2d stages and their autograd wrappers.

One autograd node per stage group (LayerNorm, Physics-Attention incl. to_out + residual, MLP incl.
residual, head) instead of the ~120 nodes per layer the reference builds; the slice
weights [B,h,N,M] are never saved — backward recomputes them from x_mid inside the kernels.
"""
from __future__ import annotations

import torch
from torch.autograd import Function

from . import ops


# ------------------------------------------------------------------------------ direct gradient accumulation
def grad_targets(params):
    """Flat-bucket gradient views of `params` if EVERY one of them is managed by a gradient bucket
    (ddp.FlatGradSync, which tags its parameters with `_pa2d_slot`), else None.  With targets the backward
    kernels add their result straight into the bucket (`accumulate = 1`) and the autograd node returns None for
    those parameters — no per-call gradient tensors, no torch `add_` launches, no copies."""
    slots = [getattr(p, "_pa2d_slot", None) for p in params]
    if not slots or any(s is None for s in slots):
        return None
    return tuple(sync.target(i) for sync, i in slots)


def _ret(targets, grads):
    """Gradients to hand back to autograd: None for everything that was accumulated in place."""
    return tuple(None for _ in grads) if targets is not None else tuple(grads)


# ------------------------------------------------------------------------------ Physics-Attention
def attn_forward(xn, P, res, H, W, heads, engine=None, xn_planes=None, shape=None):
    """xn [B,N,C] (already layer-normed).  P: dict of parameter tensors.  Returns (out, saved).
    H is None -> irregular-mesh variant (Physics_Attention.py:6-57): Linear projections, no
    temperature clamp; otherwise the structured-mesh variant (3x3 conv projections, clamp).
    xn_planes (structured, bf16 engines): the LayerNorm output exists only as the conv's bf16 plane image
    (ops.layernorm_fwd_planes); `xn` is then None and `shape` = (B, N, C)."""
    B, N, C = xn.shape if xn is not None else shape
    D = C // heads
    M = P["ws"].shape[0]
    temp = P["temperature"].reshape(heads).contiguous()
    structured = H is not None
    if xn_planes is not None:
        xf = ops.conv3x3x2_fwd_planes(xn_planes, P["wx"], P["bx"], P["wf"], P["bf"], B, H, W, engine)
    elif structured:
        xf = ops.conv3x3x2_fwd(xn, P["wx"], P["bx"], P["wf"], P["bf"], H, W, engine=engine)    # [B,N,2C]
    else:   # both Linear(C, C) projections as ONE GEMM with the weights stacked along the output dim
        wcat, bcat = torch.cat((P["wx"], P["wf"]), 0), torch.cat((P["bx"], P["bf"]), 0)
        xf = ops.linear_fwd(xn.view(B * N, C), wcat, bcat, engine=engine)[0].view(B, N, 2 * C)
    spart, npart = ops.slice_scatter(xf, 2 * C, 0, xf, 2 * C, C, P["ws"], P["bs"], temp, B, N, heads, D, M,
                                     clamp=structured, engine=engine)
    s, nrm, o = ops.token_attn_fwd(spart, npart, P["wq"], P["wk"], P["wv"])
    y = ops.deslice_fwd(xf, 2 * C, 0, o, P["ws"], P["bs"], temp, B, N, heads, D, M, clamp=structured, engine=engine)  # [B,N,C]
    out, _ = ops.linear_fwd(y.view(B * N, C), P["wo"], P["bo"],
                            res=None if res is None else res.reshape(B * N, C), engine=engine)
    return out.view(B, N, C), (xn if xn_planes is None else xn_planes, xf, s, nrm, o, y, temp)


def attn_backward(saved, P, dout, H, W, heads, need_dx=True, engine=None, targets=None, planes=False):
    """`targets`: dict ATTN_KEYS -> gradient buffer to ADD into (structured meshes), or None (fresh tensors).
    `planes`: saved[0] is the plane image of the LayerNorm output; the slice backward then emits [dX | dF] as planes too
    (plus the conv bias gradients) and the conv backward consumes both images directly."""
    xn, xf, s, nrm, o, y, temp = saved
    B, N, C2 = xf.shape
    C = C2 // 2
    D = C // heads
    M = P["ws"].shape[0]
    d2 = dout.reshape(B * N, C)
    structured = H is not None
    T = targets if structured else None
    t = (lambda *ks: tuple(T[k] for k in ks)) if T is not None else (lambda *ks: None)
    dy = ops.linear_bwd_data(d2, P["wo"], engine=engine)                                      # [B*N,C]
    dwo, dbo = ops.linear_bwd_weight(d2, y.view(B * N, C), engine=engine, into=t("wo", "bo"))
    dopart, _ = ops.slice_scatter(xf, 2 * C, 0, dy, C, 0, P["ws"], P["bs"], temp, B, N, heads, D, M,
                                  want_norm=False, clamp=structured, engine=engine)
    ds, dn, dwq, dwk, dwv = ops.token_attn_bwd(s, nrm, P["wq"], P["wk"], P["wv"], dopart, into=t("wq", "wk", "wv"))
    if planes:
        dxfp, dbx, dbf, dws, dbs, dtemp = ops.slice_bwd_points_planes(xf, dy, P["ws"], P["bs"], temp, o, ds, dn, nrm, B, N, heads,
                                                                      D, M, engine, clamp=True,
                                                                      into=t("bx", "bf", "ws", "bs", "temperature"))
        dxn, dwx, dwf = ops.conv3x3x2_bwd_planes(dxfp, xn, P["wx"], P["wf"], B, H, W, engine, need_dx=need_dx,
                                                 into=t("wx", "wf"))
        if T is not None:
            return dxn, None
        return dxn, dict(temperature=dtemp.view(1, heads, 1, 1), wx=dwx, bx=dbx, wf=dwf, bf=dbf, ws=dws, bs=dbs,
                         wq=dwq, wk=dwk, wv=dwv, wo=dwo, bo=dbo)
    dxf, dws, dbs, dtemp = ops.slice_bwd_points(xf, dy, P["ws"], P["bs"], temp, o, ds, dn, B, N, heads, D, M,
                                                clamp=structured, into=t("ws", "bs", "temperature"), engine=engine)
    if structured:
        dxn, dwx, dbx, dwf, dbf = ops.conv3x3x2_bwd(dxf, xn, P["wx"], P["wf"], H, W, need_dx=need_dx, engine=engine,
                                                    into=t("wx", "bx", "wf", "bf"))
    else:
        dxf2, xn2 = dxf.view(B * N, 2 * C), xn.view(B * N, C)
        dwcat, dbcat = ops.linear_bwd_weight(dxf2, xn2, engine=engine)
        dwx, dwf, dbx, dbf = dwcat[:C].contiguous(), dwcat[C:].contiguous(), dbcat[:C].contiguous(), dbcat[C:].contiguous()
        dxn = ops.linear_bwd_data(dxf2, torch.cat((P["wx"], P["wf"]), 0), engine=engine).view(B, N, C) if need_dx else None
    if T is not None:
        return dxn, None
    grads = dict(temperature=dtemp.view(1, heads, 1, 1), wx=dwx, bx=dbx, wf=dwf, bf=dbf, ws=dws, bs=dbs,
                 wq=dwq, wk=dwk, wv=dwv, wo=dwo, bo=dbo)
    return dxn, grads


def _attn_targets(params, structured):
    if not structured:
        return None
    tg = grad_targets(params)
    return None if tg is None else dict(zip(ATTN_KEYS, tg))


def _attn_grads(g):
    return tuple(None for _ in ATTN_KEYS) if g is None else tuple(g[k] for k in ATTN_KEYS)


ATTN_KEYS = ("temperature", "wx", "bx", "wf", "bf", "ws", "bs", "wq", "wk", "wv", "wo", "bo")


class PhysicsAttentionFn(Function):
    """out = to_out(deslice(attn(slice(conv(xn))))) (+ res)."""

    @staticmethod
    def forward(ctx, xn, res, H, W, heads, engine, *params):
        P = dict(zip(ATTN_KEYS, (p.detach().contiguous() for p in params)))
        out, saved = attn_forward(xn.detach().contiguous(), P, None if res is None else res.detach().contiguous(),
                                  H, W, heads, engine)
        ctx.P, ctx.saved, ctx.geom, ctx.params = P, saved, (H, W, heads, engine), params
        ctx.has_res = res is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        H, W, heads, engine = ctx.geom
        dout = dout.contiguous()
        dxn, g = attn_backward(ctx.saved, ctx.P, dout, H, W, heads, need_dx=ctx.needs_input_grad[0], engine=engine,
                               targets=_attn_targets(ctx.params, H is not None))
        return (dxn, dout if ctx.has_res else None, None, None, None, None) + _attn_grads(g)


# ------------------------------------------------------------------------------ MLP (Linear-act-Linear)
def mlp_forward(x2d, w1, b1, w2, b2, act, res2d, need_bwd=True, engine=None):
    """need_bwd=False (inference): the pre-activation is not written (one [rows, r*C] store less)."""
    # what is saved is act'(pre-activation): the only use of the pre-activation in the backward (for GELU every term of
    # the derivative is already computed with the activation; the data-gradient epilogue becomes one multiplication)
    hact, hpre = ops.linear_fwd(x2d, w1, b1, act=act, want_pre=need_bwd, engine=engine, save_derivative=True)
    out, _ = ops.linear_fwd(hact, w2, b2, res=res2d, engine=engine)
    return out, (x2d, hpre, hact)


def mlp_backward(saved, w1, w2, act, dout2d, need_dx=True, engine=None, targets=None):
    """`targets` = (dw1, db1, dw2, db2) buffers to add into, or None."""
    x2d, hpre, hact = saved
    t1, t2 = (None, None) if targets is None else (targets[0:2], targets[2:4])
    dhpre = ops.linear_bwd_data(dout2d, w2, pre=hpre, act=act, engine=engine, pre_is_derivative=True)
    dw2, db2 = ops.linear_bwd_weight(dout2d, hact, engine=engine, into=t2)
    dw1, db1 = ops.linear_bwd_weight(dhpre, x2d, engine=engine, into=t1)
    dx = ops.linear_bwd_data(dhpre, w1, engine=engine) if need_dx else None
    return dx, dw1, db1, dw2, db2


class MLPFn(Function):
    """linear_post(act(linear_pre(x))) (+ res) for the n_layers=0 MLP of the reference."""

    @staticmethod
    def forward(ctx, x, res, act, engine, w1, b1, w2, b2):
        shp = x.shape
        ctx.params = (w1, b1, w2, b2)
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        w1, b1, w2, b2 = (t.detach().contiguous() for t in (w1, b1, w2, b2))
        res2d = None if res is None else res.detach().reshape(-1, w2.shape[0]).contiguous()
        out, saved = mlp_forward(x2d, w1, b1, w2, b2, act, res2d, need_bwd=any(ctx.needs_input_grad), engine=engine)
        ctx.saved, ctx.w, ctx.act, ctx.shp, ctx.has_res, ctx.engine = saved, (w1, w2), act, shp, res is not None, engine
        return out.view(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, dout):
        w1, w2 = ctx.w
        d2 = dout.reshape(-1, w2.shape[0]).contiguous()
        tg = grad_targets(ctx.params)     # None when w1 is the zero-padded copy of a parameter: autograd handles it
        dx, *gw = mlp_backward(ctx.saved, w1, w2, ctx.act, d2, need_dx=ctx.needs_input_grad[0], engine=ctx.engine,
                               targets=tg)
        return (None if dx is None else dx.view(ctx.shp), dout if ctx.has_res else None, None, None) + _ret(tg, gw)


# ------------------------------------------------------------------------------ LayerNorm / head
class LayerNormFn(Function):
    @staticmethod
    def forward(ctx, x, gamma, beta):
        shp = x.shape
        ctx.params = (gamma, beta)
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        gamma, beta = gamma.detach().contiguous(), beta.detach().contiguous()
        y, mean, rstd = ops.layernorm_fwd(x2d, gamma, beta)
        ctx.saved, ctx.shp = (x2d, mean, rstd, gamma), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        x2d, mean, rstd, gamma = ctx.saved
        tg = grad_targets(ctx.params)
        dx, dg, db = ops.layernorm_bwd(dy.reshape(x2d.shape).contiguous(), x2d, mean, rstd, gamma, into=tg)
        return (dx.view(ctx.shp),) + _ret(tg, (dg, db))


class HeadFn(Function):
    """mlp2: Linear(C, out_dim <= 8)."""

    @staticmethod
    def forward(ctx, xn, w, b):
        shp = xn.shape
        ctx.params = (w, b)
        x2d = xn.detach().reshape(-1, shp[-1]).contiguous()
        w, b = w.detach().contiguous(), b.detach().contiguous()
        ctx.saved, ctx.shp = (x2d, w), shp
        return ops.head_fwd(x2d, w, b).view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, w = ctx.saved
        tg = grad_targets(ctx.params)
        dxn, dw, db = ops.head_bwd(dy.reshape(-1, w.shape[0]).contiguous(), x2d, w, into=tg)
        return (dxn.view(ctx.shp),) + _ret(tg, (dw, db))


class AttnBranchFn(Function):
    """fx + Attn(LayerNorm(fx)) as ONE autograd node (Transolver_block.forward, …_2D.py:70): the residual
    gradient is folded into the LayerNorm backward kernel (`dres`), so no separate elementwise add runs."""

    @staticmethod
    def forward(ctx, fx, ln_w, ln_b, H, W, heads, engine, *params):
        shp = fx.shape
        ctx.params, ctx.ln_params = params, (ln_w, ln_b)
        fx2d = fx.detach().reshape(-1, shp[-1]).contiguous()
        ln_w, ln_b = ln_w.detach().contiguous(), ln_b.detach().contiguous()
        P = dict(zip(ATTN_KEYS, (p.detach().contiguous() for p in params)))
        # bf16 engines on fp32 storage: LayerNorm writes the conv's operand planes directly (no fp32 xn, no pre-pass)
        ctx.planes = (H is not None and fx2d.dtype == torch.float32 and len(shp) == 3
                      and ops.conv_planes_mask(shp[0], H, W, shp[2], engine) == 7)
        if ctx.planes:
            xnp, mean, rstd = ops.layernorm_fwd_planes(fx2d, ln_w, ln_b, engine)
            out, saved = attn_forward(None, P, fx2d.view(shp), H, W, heads, engine, xn_planes=xnp, shape=tuple(shp))
        else:
            xn, mean, rstd = ops.layernorm_fwd(fx2d, ln_w, ln_b)
            out, saved = attn_forward(xn.view(shp), P, fx2d.view(shp), H, W, heads, engine)
        ctx.P, ctx.saved, ctx.geom, ctx.ln = P, saved, (H, W, heads, engine), (fx2d, mean, rstd, ln_w)
        return out

    @staticmethod
    def backward(ctx, dout):
        H, W, heads, engine = ctx.geom
        fx2d, mean, rstd, ln_w = ctx.ln
        dout = dout.contiguous()
        dxn, g = attn_backward(ctx.saved, ctx.P, dout, H, W, heads, need_dx=True, engine=engine,
                               targets=_attn_targets(ctx.params, H is not None), planes=ctx.planes)
        tl = grad_targets(ctx.ln_params)
        dfx, dg, db = ops.layernorm_bwd(dxn.reshape(fx2d.shape), fx2d, mean, rstd, ln_w, dres=dout.reshape(fx2d.shape),
                                        into=tl)
        return (dfx.view(dout.shape),) + _ret(tl, (dg, db)) + (None, None, None, None) + _attn_grads(g)


class MLPBranchFn(Function):
    """fx + MLP(LayerNorm(fx)) as one autograd node (…_2D.py:71), residual gradient folded into LN backward."""

    @staticmethod
    def forward(ctx, fx, ln_w, ln_b, act, engine, w1, b1, w2, b2):
        shp = fx.shape
        ctx.params, ctx.ln_params = (w1, b1, w2, b2), (ln_w, ln_b)
        fx2d = fx.detach().reshape(-1, shp[-1]).contiguous()
        ln_w, ln_b, w1, b1, w2, b2 = (t.detach().contiguous() for t in (ln_w, ln_b, w1, b1, w2, b2))
        xn, mean, rstd = ops.layernorm_fwd(fx2d, ln_w, ln_b)
        out, saved = mlp_forward(xn, w1, b1, w2, b2, act, fx2d, need_bwd=any(ctx.needs_input_grad), engine=engine)
        ctx.saved, ctx.w, ctx.act, ctx.ln, ctx.engine = saved, (w1, w2), act, (fx2d, mean, rstd, ln_w), engine
        return out.view(shp)

    @staticmethod
    def backward(ctx, dout):
        w1, w2 = ctx.w
        fx2d, mean, rstd, ln_w = ctx.ln
        d2 = dout.reshape(fx2d.shape).contiguous()
        tg, tl = grad_targets(ctx.params), grad_targets(ctx.ln_params)
        dxn, *gw = mlp_backward(ctx.saved, w1, w2, ctx.act, d2, need_dx=True, engine=ctx.engine, targets=tg)
        dfx, dg, db = ops.layernorm_bwd(dxn, fx2d, mean, rstd, ln_w, dres=d2, into=tl)
        return (dfx.view(dout.shape),) + _ret(tl, (dg, db)) + (None, None) + _ret(tg, gw)


def attn_branch(fx, ln_w, ln_b, H, W, heads, params, engine=None):
    return AttnBranchFn.apply(fx, ln_w, ln_b, H, W, heads, engine, *params)


def mlp_branch(fx, ln_w, ln_b, act, w1, b1, w2, b2, engine=None):
    return MLPBranchFn.apply(fx, ln_w, ln_b, act, engine, w1, b1, w2, b2)


class LinearFn(Function):
    """y = act(x . w^T + b): generic dense layer (MLP hidden layers when n_layers > 0, wide heads)."""

    @staticmethod
    def forward(ctx, x, act, engine, w, b):
        shp = x.shape
        x2d = x.detach().reshape(-1, shp[-1]).contiguous()
        w = w.detach().contiguous()
        b = None if b is None else b.detach().contiguous()
        y, pre = ops.linear_fwd(x2d, w, b, act=act, want_pre=act is not None, engine=engine)
        ctx.saved, ctx.act, ctx.shp, ctx.has_b, ctx.engine = (x2d, w, pre), act, shp, b is not None, engine
        return y.view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, w, pre = ctx.saved
        d2 = dy.reshape(-1, w.shape[0]).contiguous()
        if ctx.act is not None:
            d2 = ops.act_bwd(d2, pre, ctx.act)
        dw, db = ops.linear_bwd_weight(d2, x2d, want_bias=ctx.has_b, engine=ctx.engine)
        dx = ops.linear_bwd_data(d2, w, engine=ctx.engine).view(ctx.shp) if ctx.needs_input_grad[0] else None
        return dx, None, None, dw, db


def linear(x, w, b, act, engine=None):
    return LinearFn.apply(x, act, engine, w, b)


def layer_norm(x, gamma, beta):
    return LayerNormFn.apply(x, gamma, beta)


def physics_attention(xn, res, H, W, heads, params, engine=None):
    return PhysicsAttentionFn.apply(xn, res, H, W, heads, engine, *params)


def mlp(x, res, act, w1, b1, w2, b2, engine=None):
    return MLPFn.apply(x, res, act, engine, w1, b1, w2, b2)


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
        # d ratio_b / d pred = (pred - y) / (dn_b * yn_b), times the per-sample upstream gradient (may be 0)
        dpred = ops.rel_l2_bwd(p2, y2, dn, yn, gratio.contiguous())
        return dpred.view(ctx.shp), None


def rel_l2(pred, y):
    return RelL2Fn.apply(pred, y)
