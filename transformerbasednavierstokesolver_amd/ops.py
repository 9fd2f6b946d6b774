"""Stage-level wrappers over the libpa2d C ABI.

PyTorch is used here only as plumbing: it owns device memory (outputs and workspaces come from the
caching allocator, so nothing is allocated inside the native calls) and provides the current HIP
stream.  Every function takes contiguous fp32 CUDA(HIP) tensors and raises otherwise — there is no
CPU path (the CPU restatement lives in oracle/ and is test infrastructure only).

Parameter-gradient outputs: every backward op takes `into=` — a tuple of existing gradient buffers (the
views of the flat gradient bucket) that the kernels ADD to (`accumulate = 1` in the C ABI); without it the
op allocates fresh tensors and overwrites them.
"""
from __future__ import annotations

import torch

from . import _lib

ACT_IDS = {None: 0, "none": 0, "gelu": 1, "tanh": 2, "sigmoid": 3, "relu": 4, "softplus": 5, "ELU": 6, "silu": 7}
LN_EPS = 1e-5


# Optional provider of (hipEvent_t start, hipEvent_t stop) handles, called as provider(kind) with kind in
# {"conv", "slice_scatter", "deslice", "slice_bwd"}; libpa2d records them on the launch stream right around that
# kernel (bench.py installs one to time the roofline kernels live); None = off.
event_provider = None


def _events(kind):
    return event_provider(kind) if event_provider is not None else (0, 0)


# GEMM engines (include/pa2d.h: enum pa2d_engine).  The engine is an explicit argument of every dense op; `None`
# means pa2d_default_engine() (env PA2D_GEMM=f32|split|bf16, else the fp32-accurate split engine).
ENGINE_F32, ENGINE_SPLIT, ENGINE_BF16 = 0, 1, 2
# bf16 STORAGE (BASELINE configs[2] / [4] as stated): not an engine id of the C ABI but a host-side mode — the blocks'
# activations, saved tensors and inter-kernel gradients are torch.bfloat16 tensors, and every op below dispatches on the
# activation dtype to the *_bf16 entry points (one-term bf16 MFMA, fp32 accumulate, fp32 parameters / statistics).
# fp32-in/fp32-out calls made by a model in this mode (the input embedding) run on ENGINE_BF16.
ENGINE_BF16S = 3
ENGINE_NAMES = {"f32": ENGINE_F32, "split": ENGINE_SPLIT, "bf16": ENGINE_BF16, "bf16s": ENGINE_BF16S}


def default_engine():
    import os
    if os.environ.get("PA2D_GEMM", "") == "bf16s":      # host-side mode (see ENGINE_BF16S), not an ABI engine
        return ENGINE_BF16S
    return _L().pa2d_default_engine()


def resolve_engine(engine):
    if engine is None:
        return default_engine()
    if isinstance(engine, str):
        return ENGINE_NAMES[engine]
    if engine not in (ENGINE_F32, ENGINE_SPLIT, ENGINE_BF16, ENGINE_BF16S):
        raise ValueError(f"unknown GEMM engine {engine!r}")
    return int(engine)


def _abi_engine(engine):
    """Engine id for an fp32-I/O entry point of the C ABI (bf16-storage models run those on the bf16-compute engine)."""
    eng = resolve_engine(engine)
    return ENGINE_BF16 if eng == ENGINE_BF16S else eng


def _L():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libpa2d ops need tensors on the GPU (no CPU fallback exists in this package)")
        if t.dtype != torch.float32:
            raise TypeError(f"libpa2d ops are fp32; got {t.dtype}")
        if not t.is_contiguous():
            raise ValueError("libpa2d ops need contiguous tensors")


def _chk_act(*ts):
    """Activation tensors: contiguous GPU fp32 or bf16, all of ONE dtype.  Returns True for bf16 storage."""
    dt = None
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("libpa2d ops need tensors on the GPU (no CPU fallback exists in this package)")
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"libpa2d activations are fp32 or bf16; got {t.dtype}")
        if dt is not None and t.dtype != dt:
            raise TypeError(f"activation tensors of one op must share a storage type; got {dt} and {t.dtype}")
        dt = t.dtype
        if not t.is_contiguous():
            raise ValueError("libpa2d ops need contiguous tensors")
    return dt == torch.bfloat16


def _p(t, offset_elems=0):
    return 0 if t is None else t.data_ptr() + t.element_size() * offset_elems


def _fn(name, bf):
    return getattr(_L(), name + "_bf16" if bf else name)


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


def _grad_outputs(into, shapes, like):
    """(tensors, accumulate flag): the caller's buffers (added to) or fresh ones (overwritten)."""
    if into is not None:
        if len(into) != len(shapes):
            raise ValueError("`into` must hold one buffer per gradient output")
        for t, shp in zip(into, shapes):
            _chk(t)
            if t.numel() != int(torch.Size(shp).numel()):
                raise ValueError("gradient buffer of the wrong size")
        return tuple(into), 1
    return tuple(torch.empty(shp, dtype=torch.float32, device=like.device) for shp in shapes), 0


# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps=LN_EPS):
    _chk(gamma, beta)
    bf = _chk_act(x2d)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    _lib.check(_fn("pa2d_layernorm_fwd", bf)(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, Cc, eps,
                                       _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x2d, mean, rstd, gamma, dres=None, into=None):
    _chk(mean, rstd, gamma)
    bf = _chk_act(dy, x2d, dres)
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    (dg, db), acc = _grad_outputs(into, (gamma.shape, gamma.shape), x2d)
    nb = _L().pa2d_layernorm_bwd_workspace(rows, Cc)
    ws = _ws(nb, x2d)
    _lib.check(_fn("pa2d_layernorm_bwd", bf)(_p(dy), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dg),
                                       _p(db), ws.data_ptr(), nb, rows, Cc, acc, _stream()), "layernorm_bwd")
    return dx, dg, db


ACT_SAVE_DERIVATIVE = 0x100      # include/pa2d.h: PA2D_ACT_SAVE_DERIVATIVE


def linear_fwd(x2d, w, bias=None, res=None, act=None, want_pre=False, engine=None, save_derivative=False):
    """y = act(x . w^T + bias) (+ res); returns (y, pre) with pre = pre-activation if want_pre — or act'(pre-activation)
    with save_derivative (what `linear_bwd_data(..., pre_is_derivative=True)` multiplies by)."""
    _chk(w, bias)
    bf = _chk_act(x2d, res)
    M, K = x2d.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=x2d.dtype, device=x2d.device)
    pre = torch.empty_like(y) if want_pre else None
    aflag = ACT_SAVE_DERIVATIVE if (save_derivative and want_pre and act is not None) else 0
    if bf:
        _lib.check(_L().pa2d_gemm_bias_act_fwd_bf16(_p(x2d), K, _p(w), w.shape[1], _p(bias), _p(res), N, _p(y), N,
                                                    _p(pre), N, M, N, K, ACT_IDS[act] | aflag, _stream()), "gemm_bias_act_fwd_bf16")
    else:
        eng = _abi_engine(engine)
        nb = _L().pa2d_gemm_fwd_workspace(N, K, eng)
        img = _lin_image(w, 0, N, K, eng, nb)              # inside weights_frozen(): made once per weight
        ws = _ws(nb, x2d) if (nb and not img) else None
        _lib.check(_L().pa2d_gemm_bias_act_fwd(_p(x2d), K, _p(w), w.shape[1], _p(bias), _p(res), N, _p(y), N, _p(pre), N,
                                               img, _p(ws), nb if ws is not None else 0, M, N, K, ACT_IDS[act] | aflag, eng,
                                               _stream()), "gemm_bias_act_fwd")
    return y, pre


def linear_bwd_data(dy, w, pre=None, act=None, engine=None, pre_is_derivative=False):
    """dx = (dy . w) * act'(pre); pre_is_derivative: `pre` already holds act'(pre-activation) (linear_fwd(save_derivative=True))."""
    _chk(w)
    bf = _chk_act(dy, pre)
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=dy.dtype, device=dy.device)
    aflag = ACT_SAVE_DERIVATIVE if (pre_is_derivative and pre is not None and act is not None) else 0
    if bf:
        wt = torch.empty(K * N, dtype=torch.float32, device=dy.device)
        _lib.check(_L().pa2d_gemm_bwd_data_bf16(_p(dy), N, _p(w), K, _p(pre), K, ACT_IDS[act] | aflag, _p(dx), K, _p(wt), M, N, K,
                                                _stream()), "gemm_bwd_data_bf16")
    else:
        eng = _abi_engine(engine)
        nimg = _L().pa2d_gemm_fwd_workspace(K, N, eng)     # the data gradient is a GEMM of output width K, contraction N
        img = _lin_image(w, 1, K, N, eng, nimg)
        nb = K * N * 4 if img else _L().pa2d_gemm_bwd_data_workspace(N, K, eng)
        ws = _ws(nb, dy)
        _lib.check(_L().pa2d_gemm_bwd_data(_p(dy), N, _p(w), K, _p(pre), K, ACT_IDS[act] | aflag, _p(dx), K, img, ws.data_ptr(), nb,
                                           M, N, K, eng, _stream()), "gemm_bwd_data")
    return dx


def linear_bwd_weight(dy, x2d, want_bias=True, engine=None, into=None):
    """dw [N,K], db [N] (None if not wanted).  `into` = (dw_buffer, db_buffer or None): accumulate."""
    bf = _chk_act(dy, x2d)
    M, N = dy.shape
    K = x2d.shape[1]
    eng = _abi_engine(engine)
    if into is not None:
        dw, db = into
        _chk(dw, db)
        if dw.numel() != N * K or (db is not None and db.numel() != N):
            raise ValueError("gradient buffer of the wrong size")
        if not want_bias:
            db = None
        acc = 1
    else:
        dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
        db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
        acc = 0
    if bf:
        nb = _L().pa2d_gemm_bwd_weight_workspace_bf16(M, N, K)
        ws = _ws(nb, dy)
        _lib.check(_L().pa2d_gemm_bwd_weight_bf16(_p(dy), N, _p(x2d), K, _p(dw), _p(db), ws.data_ptr(), nb, M, N, K, acc,
                                                  _stream()), "gemm_bwd_weight_bf16")
        return dw, db
    nb = _L().pa2d_gemm_bwd_weight_workspace(M, N, K, eng)
    ws = _ws(nb, dy)
    _lib.check(_L().pa2d_gemm_bwd_weight(_p(dy), N, _p(x2d), K, _p(dw), _p(db), ws.data_ptr(), nb, M, N, K, acc, eng,
                                         _stream()), "gemm_bwd_weight")
    return dw, db


# ---------------------------------------------------------------------------------------------- conv weight packs
class _FrozenScope:
    """Book-keeping of one `weights_frozen()` scope: packs made inside it, keyed by (weights, dims, direction,
    engine) — the pack layout depends on the engine, which is part of the key, so models on different engines
    can share a scope."""

    def __init__(self):
        self.packs = {}
        self.images = {}      # weight plane images of the row-stationary linear kernel: (w ptr, transposed, N, K, engine)

    def refresh(self):
        """Re-pack every entry in place (same device pointers): called before replaying a hipGraph that was captured
        inside this scope, so the graph's conv launches always see the current weights."""
        for (_, _, B, H, W, Cc, direction, eng), (wx, wf, pack) in self.packs.items():
            _make_pack(wx, wf, pack, B, H, W, Cc, direction, eng)
        for (_, transposed, N, K, eng), (w, img) in self.images.items():
            _make_image(w, transposed, img, N, K, eng)


_frozen = []      # stack of active scopes


class weights_frozen:
    """Context manager declaring that no parameter changes inside it (the model calls + backward of one training
    iteration, a rollout).  Inside, the K-major pack of each layer's conv weights is built on first use and
    reused by later calls; outside a scope every conv call packs its weights itself (always correct, whatever
    changed the weights).  `with weights_frozen() as scope:` exposes `scope.refresh()`."""

    def __enter__(self):
        _frozen.append(_FrozenScope())
        return _frozen[-1]

    def __exit__(self, *exc):
        _frozen.pop()
        return False


def _make_image(w, transposed, img, N, K, eng):
    _lib.check(_L().pa2d_gemm_weight_image(_p(w), w.shape[1], transposed, img.data_ptr(), img.numel(), N, K, eng, _stream()),
               "gemm_weight_image")


def _lin_image(w, transposed, N, K, eng, nbytes):
    """Image pointer for the active `weights_frozen()` scope (0 = let the GEMM call make it in its scratch): N, K are
    the GEMM's output width and contraction length (forward: the layer's [N, K]; data gradient: [K, N])."""
    if not _frozen or not nbytes:
        return 0
    scope = _frozen[-1]
    key = (w.data_ptr(), transposed, N, K, eng)
    hit = scope.images.get(key)
    if hit is None:
        img = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
        _make_image(w, transposed, img, N, K, eng)
        hit = scope.images[key] = (w, img)
    return hit[1].data_ptr()


def _make_pack(wx, wf, pack, B, H, W, Cc, direction, eng):
    """eng = ENGINE_BF16S: the pack of the bf16-storage conv entry points (always the bf16 kernels' layout)."""
    if eng == ENGINE_BF16S:
        _lib.check(_L().pa2d_conv3x3x2_pack_bf16(_p(wx), _p(wf), pack.data_ptr(), pack.numel(), Cc, direction, _stream()),
                   "conv3x3x2_pack_bf16")
    else:
        _lib.check(_L().pa2d_conv3x3x2_pack(_p(wx), _p(wf), pack.data_ptr(), pack.numel(), B, H, W, Cc, direction, eng,
                                            _stream()), "conv3x3x2_pack")


def _conv_pack(wx, wf, B, H, W, Cc, direction, eng):
    """Pack pointer for the active scope (0 = let the conv call pack into its workspace)."""
    if not _frozen:
        return 0
    scope = _frozen[-1]
    key = (wx.data_ptr(), wf.data_ptr(), B, H, W, Cc, direction, eng)
    hit = scope.packs.get(key)
    if hit is None:
        nb = _L().pa2d_conv3x3x2_pack_bytes(Cc)
        pack = torch.empty(nb, dtype=torch.uint8, device=wx.device)
        _make_pack(wx, wf, pack, B, H, W, Cc, direction, eng)
        hit = scope.packs[key] = (wx, wf, pack)
    return hit[2].data_ptr()


def conv3x3x2_fwd(xn, wx, bx, wf, bf, H, W, engine=None):
    """xn [B,N,C] -> [B,N,2C] = [x_mid | fx_mid]."""
    _chk(wx, bx, wf, bf)
    b16 = _chk_act(xn)
    B, N, Cc = xn.shape
    eng = ENGINE_BF16S if b16 else _abi_engine(engine)
    out = torch.empty(B, N, 2 * Cc, dtype=xn.dtype, device=xn.device)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 0, eng)
    e0, e1 = _events("conv")
    if b16:
        nb = _L().pa2d_conv3x3x2_fwd_workspace_bf16(B, H, W, Cc)
        ws = _ws(nb, xn)
        _lib.check(_L().pa2d_conv3x3x2_fwd_bf16(_p(xn), _p(wx), _p(bx), _p(wf), _p(bf), _p(out), pre, ws.data_ptr(), nb,
                                                B, H, W, Cc, _stream(), e0, e1), "conv3x3x2_fwd_bf16")
        return out
    nb = _L().pa2d_conv3x3x2_fwd_workspace(B, H, W, Cc, eng)
    ws = _ws(nb, xn)
    _lib.check(_L().pa2d_conv3x3x2_fwd(_p(xn), _p(wx), _p(bx), _p(wf), _p(bf), _p(out), pre, ws.data_ptr(), nb,
                                       B, H, W, Cc, eng, _stream(), e0, e1), "conv3x3x2_fwd")
    return out


def conv3x3x2_bwd(dout, xn, wx, wf, H, W, need_dx=True, engine=None, into=None):
    """Returns (dxn, dwx, dbx, dwf, dbf); `into` = (dwx, dbx, dwf, dbf) buffers to accumulate into."""
    _chk(wx, wf)
    b16 = _chk_act(dout, xn)
    B, N, Cc = xn.shape
    eng = ENGINE_BF16S if b16 else _abi_engine(engine)
    dxn = torch.empty_like(xn) if need_dx else None
    (dwx, dbx, dwf, dbf), acc = _grad_outputs(into, (wx.shape, (Cc,), wf.shape, (Cc,)), xn)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 1, eng) if need_dx else 0
    e0, e1 = _events("conv") if need_dx else (0, 0)
    if b16:
        nb = _L().pa2d_conv3x3x2_workspace_bf16(B, H, W, Cc)
        ws = _ws(nb, xn)
        _lib.check(_L().pa2d_conv3x3x2_bwd_bf16(_p(dout), _p(xn), _p(wx), _p(wf), _p(dxn), _p(dwx), _p(dbx), _p(dwf),
                                                _p(dbf), pre, ws.data_ptr(), nb, B, H, W, Cc, acc, _stream(), e0, e1),
                   "conv3x3x2_bwd_bf16")
        return dxn, dwx, dbx, dwf, dbf
    nb = _L().pa2d_conv3x3x2_workspace(B, H, W, Cc, eng)
    ws = _ws(nb, xn)
    _lib.check(_L().pa2d_conv3x3x2_bwd(_p(dout), _p(xn), _p(wx), _p(wf), _p(dxn), _p(dwx), _p(dbx), _p(dwf), _p(dbf),
                                       pre, ws.data_ptr(), nb, B, H, W, Cc, acc, eng, _stream(), e0, e1),
               "conv3x3x2_bwd")
    return dxn, dwx, dbx, dwf, dbf


# ---------------------------------------------------------------------------------------------- operand planes
def conv_planes_mask(B, H, W, Cc, engine):
    """7 when `engine` consumes every conv operand at this shape as bf16 planes (then LayerNorm / slice backward can emit
    the planes directly: layernorm_fwd_planes, slice_bwd_points_planes, conv3x3x2_fwd/bwd_planes)."""
    eng = _abi_engine(engine)
    return _L().pa2d_conv3x3x2_planes_mask(B, H, W, Cc, eng) if eng in (ENGINE_SPLIT, ENGINE_BF16) else 0


def _planes(rows, Cc, eng, like):
    return torch.empty(_L().pa2d_planes_bytes(rows, Cc, eng), dtype=torch.uint8, device=like.device)


def layernorm_fwd_planes(x2d, gamma, beta, engine, eps=LN_EPS):
    """LayerNorm whose output exists ONLY as the bf16 plane image the conv GEMMs stage.  Returns (planes, mean, rstd)."""
    _chk(x2d, gamma, beta)
    rows, Cc = x2d.shape
    eng = _abi_engine(engine)
    planes = _planes(rows, Cc, eng, x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    _lib.check(_L().pa2d_layernorm_fwd_planes(_p(x2d), _p(gamma), _p(beta), planes.data_ptr(), _p(mean), _p(rstd), rows,
                                              Cc, eps, eng, _stream()), "layernorm_fwd_planes")
    return planes, mean, rstd


def conv3x3x2_fwd_planes(xn_planes, wx, bx, wf, bf, B, H, W, engine):
    """[x_mid | fx_mid] [B, H*W, 2C] fp32 from the plane image of the LayerNorm output."""
    _chk(wx, bx, wf, bf)
    Cc = wx.shape[0]
    eng = _abi_engine(engine)
    out = torch.empty(B, H * W, 2 * Cc, dtype=torch.float32, device=wx.device)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 0, eng)
    nb = _L().pa2d_conv3x3x2_pack_bytes(Cc)
    ws = _ws(nb, wx)
    e0, e1 = _events("conv")
    _lib.check(_L().pa2d_conv3x3x2_fwd_planes(xn_planes.data_ptr(), _p(wx), _p(bx), _p(wf), _p(bf), _p(out), pre,
                                              ws.data_ptr(), nb, B, H, W, Cc, eng, _stream(), e0, e1),
               "conv3x3x2_fwd_planes")
    return out


def conv3x3x2_bwd_planes(dout_planes, xn_planes, wx, wf, B, H, W, engine, need_dx=True, into=None):
    """(dxn, dwx, dwf) from plane images of dOut and X; `into` = (dwx, dwf) buffers to accumulate into."""
    _chk(wx, wf)
    Cc = wx.shape[0]
    eng = _abi_engine(engine)
    dxn = torch.empty(B, H * W, Cc, dtype=torch.float32, device=wx.device) if need_dx else None
    (dwx, dwf), acc = _grad_outputs(into, (wx.shape, wf.shape), wx)
    nb = _L().pa2d_conv3x3x2_workspace_planes(B, H, W, Cc, eng)
    ws = _ws(nb, wx)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 1, eng) if need_dx else 0
    e0, e1 = _events("conv") if need_dx else (0, 0)
    _lib.check(_L().pa2d_conv3x3x2_bwd_planes(dout_planes.data_ptr(), xn_planes.data_ptr(), _p(wx), _p(wf), _p(dxn), _p(dwx),
                                              _p(dwf), pre, ws.data_ptr(), nb, B, H, W, Cc, acc, eng, _stream(), e0, e1),
               "conv3x3x2_bwd_planes")
    return dxn, dwx, dwf


def slice_bwd_points_planes(xf, dy, ws_w, bs, temperature, o, ds, dn, nrm, B, N, heads, D, M, engine, clamp=True, into=None):
    """slice_bwd_points whose [dX | dF] leaves only as the plane image for the conv backward; also returns the conv bias
    gradients (`nrm` = the forward's slice norms [B*heads, M], from token_attn_fwd).
    Returns (dxf_planes, dbx, dbf, dws, dbs, dtemperature); `into` = (dbx, dbf, dws, dbs, dtemperature)."""
    _chk(xf, dy, ws_w, bs, temperature, o, ds, dn, nrm)
    Cc = heads * D
    eng = _abi_engine(engine)
    planes = _planes(B * N, 2 * Cc, eng, xf)
    (dbx, dbf, dws, dbs, dtemp), acc = _grad_outputs(into, ((Cc,), (Cc,), ws_w.shape, bs.shape, (heads,)), xf)
    nb = _L().pa2d_slice_bwd_workspace(B, N, heads, D, M)
    ws = _ws(nb, xf)
    e0, e1 = _events("slice_bwd")
    _lib.check(_L().pa2d_slice_bwd_points_planes(_p(xf), 2 * Cc, _p(xf, Cc), 2 * Cc, _p(dy), Cc, _p(ws_w), _p(bs),
                                                 _p(temperature), _p(o), _p(ds), _p(dn), _p(nrm), planes.data_ptr(), _p(dbx), _p(dbf),
                                                 _p(dws), _p(dbs), _p(dtemp), ws.data_ptr(), nb, B, N, heads, D, M,
                                                 int(clamp), acc, eng, _stream(), e0, e1), "slice_bwd_points_planes")
    return planes, dbx, dbf, dws, dbs, dtemp


def slice_nchunk(B, N, heads):
    return _L().pa2d_slice_nchunk(B, N, heads)


def _slice_engine_args(engine, bf):
    """The slice entry points of fp32 storage take the engine (exact-fp32 MFMA kernels vs bf16 splits); bf16 storage has none."""
    return () if bf else (_abi_engine(engine),)


def slice_scatter(xm, ldx, xm_off, v, ldv, v_off, ws_w, bs, temperature, B, N, heads, D, M, want_norm=True,
                  clamp=True, engine=None):
    """Partial sums of W^T V per point chunk.  xm / v are base tensors; *_off are float offsets."""
    _chk(ws_w, bs, temperature)
    bf = _chk_act(xm, v)
    nchunk = slice_nchunk(B, N, heads)
    spart = torch.empty(B * heads, nchunk, M, D, dtype=torch.float32, device=xm.device)
    npart = torch.empty(B * heads, nchunk, M, dtype=torch.float32, device=xm.device) if want_norm else None
    e0, e1 = _events("slice_scatter")
    _lib.check(_fn("pa2d_slice_scatter", bf)(_p(xm, xm_off), ldx, _p(v, v_off), ldv, _p(ws_w), _p(bs), _p(temperature),
                                       _p(spart), _p(npart), B, N, heads, D, M, int(clamp), *_slice_engine_args(engine, bf),
                                       _stream(), e0, e1),
               "slice_scatter")
    return spart, npart


def token_attn_fwd(spart, npart, wq, wk, wv):
    _chk(spart, npart, wq, wk, wv)
    BH, nchunk, M, D = spart.shape
    s = torch.empty(BH, M, D, dtype=torch.float32, device=spart.device)
    nrm = torch.empty(BH, M, dtype=torch.float32, device=spart.device)
    o = torch.empty_like(s)
    _lib.check(_L().pa2d_token_attn_fwd(_p(spart), _p(npart), _p(wq), _p(wk), _p(wv), _p(s), _p(nrm), _p(o), BH,
                                        nchunk, M, D, _stream()), "token_attn_fwd")
    return s, nrm, o


def token_attn_bwd(s, nrm, wq, wk, wv, dopart, into=None):
    _chk(s, nrm, wq, wk, wv, dopart)
    BH, nchunk, M, D = dopart.shape
    ds = torch.empty_like(s)
    dn = torch.empty_like(nrm)
    (dwq, dwk, dwv), acc = _grad_outputs(into, ((D, D), (D, D), (D, D)), s)
    nb = _L().pa2d_token_attn_bwd_workspace(BH, D)
    ws = _ws(nb, s)
    _lib.check(_L().pa2d_token_attn_bwd(_p(s), _p(nrm), _p(wq), _p(wk), _p(wv), _p(dopart), _p(ds), _p(dn), _p(dwq),
                                        _p(dwk), _p(dwv), ws.data_ptr(), nb, BH, nchunk, M, D, acc, _stream()),
               "token_attn_bwd")
    return ds, dn, dwq, dwk, dwv


def deslice_fwd(xm, ldx, xm_off, o, ws_w, bs, temperature, B, N, heads, D, M, clamp=True, engine=None):
    _chk(o, ws_w, bs, temperature)
    bf = _chk_act(xm)
    y = torch.empty(B, N, heads * D, dtype=xm.dtype, device=xm.device)
    e0, e1 = _events("deslice")
    _lib.check(_fn("pa2d_deslice_fwd", bf)(_p(xm, xm_off), ldx, _p(o), _p(ws_w), _p(bs), _p(temperature), _p(y), heads * D,
                                     B, N, heads, D, M, int(clamp), *_slice_engine_args(engine, bf), _stream(), e0, e1),
               "deslice_fwd")
    return y


def slice_bwd_points(xf, dy, ws_w, bs, temperature, o, ds, dn, B, N, heads, D, M, clamp=True, into=None, engine=None):
    """xf = [B,N,2C] ([x_mid | fx_mid]); returns dxf [B,N,2C], dws, dbs, dtemperature [heads]."""
    _chk(ws_w, bs, temperature, o, ds, dn)
    bf = _chk_act(xf, dy)
    Cc = heads * D
    dxf = torch.empty_like(xf)
    (dws, dbs, dtemp), acc = _grad_outputs(into, (ws_w.shape, bs.shape, (heads,)), xf)
    nb = _L().pa2d_slice_bwd_workspace(B, N, heads, D, M)
    ws = _ws(nb, xf)
    e0, e1 = _events("slice_bwd")
    _lib.check(_fn("pa2d_slice_bwd_points", bf)(_p(xf), 2 * Cc, _p(xf, Cc), 2 * Cc, _p(dy), Cc, _p(ws_w), _p(bs),
                                          _p(temperature), _p(o), _p(ds), _p(dn), _p(dxf), 2 * Cc, _p(dxf, Cc),
                                          2 * Cc, _p(dws), _p(dbs), _p(dtemp), ws.data_ptr(), nb, B, N, heads, D, M,
                                          int(clamp), acc, *_slice_engine_args(engine, bf), _stream(), e0, e1),
               "slice_bwd_points")
    return dxf, dws, dbs, dtemp


def head_fwd(xn2d, w, b):
    """y is always fp32 (the model output), whatever the storage type of the activations."""
    _chk(w, b)
    bf = _chk_act(xn2d)
    rows, Cc = xn2d.shape
    O = w.shape[0]
    y = torch.empty(rows, O, dtype=torch.float32, device=xn2d.device)
    _lib.check(_fn("pa2d_head_fwd", bf)(_p(xn2d), _p(w), _p(b), _p(y), rows, Cc, O, _stream()), "head_fwd")
    return y


def head_bwd(dy, xn2d, w, into=None):
    _chk(dy, w)
    bf = _chk_act(xn2d)
    rows, Cc = xn2d.shape
    O = w.shape[0]
    dxn = torch.empty_like(xn2d)
    (dw, db), acc = _grad_outputs(into, (w.shape, (O,)), w)
    nb = _L().pa2d_head_bwd_workspace(rows, Cc, O)
    ws = _ws(nb, dy)
    _lib.check(_fn("pa2d_head_bwd", bf)(_p(dy), _p(xn2d), _p(w), _p(dxn), _p(dw), _p(db), ws.data_ptr(), nb, rows, Cc, O,
                                  acc, _stream()), "head_bwd")
    return dxn, dw, db


def act_bwd(dy, pre, act):
    _chk(dy, pre)
    out = torch.empty_like(dy)
    _lib.check(_L().pa2d_act_bwd(_p(dy), _p(pre), _p(out), dy.numel(), ACT_IDS[act], _stream()), "act_bwd")
    return out


# ---------------------------------------------------------------------------------------------- SURVEY 8(f)-1
def sumsq(flat):
    _chk(flat)
    out = torch.empty(1, dtype=torch.float32, device=flat.device)
    nb = _L().pa2d_sumsq_workspace(flat.numel())
    ws = _ws(nb, flat)
    _lib.check(_L().pa2d_sumsq(_p(flat), flat.numel(), _p(out), ws.data_ptr(), nb, _stream()), "sumsq")
    return out


def adamw_step(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step_index, gnorm_sq=None, max_norm=0.0):
    _chk(p, g, m, v, gnorm_sq)
    _lib.check(_L().pa2d_adamw_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay,
                                    step_index, _p(gnorm_sq), max_norm, _stream()), "adamw_step")


def rel_l2_fwd(pred2d, y2d):
    _chk(pred2d, y2d)
    B, L = pred2d.shape
    dn = torch.empty(B, dtype=torch.float32, device=pred2d.device)
    yn, ratio = torch.empty_like(dn), torch.empty_like(dn)
    _lib.check(_L().pa2d_rel_l2_fwd(_p(pred2d), _p(y2d), _p(dn), _p(yn), _p(ratio), B, L, _stream()), "rel_l2_fwd")
    return dn, yn, ratio


def rel_l2_bwd(pred2d, y2d, dn, yn, gout):
    """gout: per-sample upstream gradient [B]."""
    _chk(pred2d, y2d, dn, yn, gout)
    B, L = pred2d.shape
    if gout.numel() != B:
        raise ValueError("gout must hold one value per sample")
    dpred = torch.empty_like(pred2d)
    _lib.check(_L().pa2d_rel_l2_bwd(_p(pred2d), _p(y2d), _p(dn), _p(yn), _p(gout), _p(dpred), B, L, _stream()),
               "rel_l2_bwd")
    return dpred
