"""Stage-level wrappers over the libpa2d C ABI.

PyTorch is used here only as plumbing: it owns device memory (outputs and workspaces come from the
caching allocator, so nothing is allocated inside the native calls) and provides the current HIP
stream.  Every function takes contiguous fp32 CUDA(HIP) tensors and raises otherwise — there is no
CPU path (the CPU restatement lives in oracle/ and is test infrastructure only).
"""
from __future__ import annotations

import torch

from . import _lib

ACT_IDS = {None: 0, "none": 0, "gelu": 1, "tanh": 2, "sigmoid": 3, "relu": 4, "softplus": 5, "ELU": 6, "silu": 7}
LN_EPS = 1e-5


# Optional provider of (hipEvent_t start, hipEvent_t stop) handles recorded around the conv
# implicit-GEMM launches (bench.py installs one to time the dominant kernel live); None = off.
conv_event_provider = None


def _conv_events():
    return conv_event_provider() if conv_event_provider is not None else (0, 0)


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


def _p(t, offset_floats=0):
    return 0 if t is None else t.data_ptr() + 4 * offset_floats


def _ws(nbytes, like):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=like.device)


# ----------------------------------------------------------------------------------------------
def layernorm_fwd(x2d, gamma, beta, eps=LN_EPS):
    _chk(x2d, gamma, beta)
    rows, Cc = x2d.shape
    y = torch.empty_like(x2d)
    mean = torch.empty(rows, dtype=torch.float32, device=x2d.device)
    rstd = torch.empty_like(mean)
    _lib.check(_L().pa2d_layernorm_fwd(_p(x2d), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), rows, Cc, eps,
                                       _stream()), "layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x2d, mean, rstd, gamma, dres=None):
    _chk(dy, x2d, mean, rstd, gamma, dres)
    rows, Cc = x2d.shape
    dx = torch.empty_like(x2d)
    dg = torch.empty_like(gamma)
    db = torch.empty_like(gamma)
    nb = _L().pa2d_layernorm_bwd_workspace(rows, Cc)
    ws = _ws(nb, x2d)
    _lib.check(_L().pa2d_layernorm_bwd(_p(dy), _p(x2d), _p(mean), _p(rstd), _p(gamma), _p(dres), _p(dx), _p(dg),
                                       _p(db), ws.data_ptr(), nb, rows, Cc, _stream()), "layernorm_bwd")
    return dx, dg, db


def linear_fwd(x2d, w, bias=None, res=None, act=None, want_pre=False):
    """y = act(x . w^T + bias) (+ res); returns (y, pre) with pre = pre-activation if want_pre."""
    _chk(x2d, w, bias, res)
    M, K = x2d.shape
    N = w.shape[0]
    y = torch.empty(M, N, dtype=torch.float32, device=x2d.device)
    pre = torch.empty_like(y) if want_pre else None
    _lib.check(_L().pa2d_gemm_bias_act_fwd(_p(x2d), K, _p(w), w.shape[1], _p(bias), _p(res), N, _p(y), N, _p(pre), N,
                                           M, N, K, ACT_IDS[act], _stream()), "gemm_bias_act_fwd")
    return y, pre


def linear_bwd_data(dy, w, pre=None, act=None):
    """dx = (dy . w) * act'(pre)."""
    _chk(dy, w, pre)
    M, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    wt = torch.empty(K * N, dtype=torch.float32, device=dy.device)
    _lib.check(_L().pa2d_gemm_bwd_data(_p(dy), N, _p(w), K, _p(pre), K, ACT_IDS[act], _p(dx), K, _p(wt), M, N, K,
                                       _stream()), "gemm_bwd_data")
    return dx


def linear_bwd_weight(dy, x2d, want_bias=True):
    _chk(dy, x2d)
    M, N = dy.shape
    K = x2d.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    nb = _L().pa2d_gemm_bwd_weight_workspace(M, N, K)
    ws = _ws(nb, dy)
    _lib.check(_L().pa2d_gemm_bwd_weight(_p(dy), N, _p(x2d), K, _p(dw), _p(db), ws.data_ptr(), nb, M, N, K,
                                         _stream()), "gemm_bwd_weight")
    return dw, db


# ---------------------------------------------------------------------------------------------- conv weight packs
class _FrozenScope:
    """Book-keeping of one `weights_frozen()` scope: packs made inside it, keyed by (weights, dims, direction)."""

    def __init__(self):
        self.packs = {}
        self.mode = _L().pa2d_get_gemm_mode()      # pack layout depends on the GEMM engine

    def refresh(self):
        """Re-pack every entry in place (same device pointers): called before replaying a hipGraph that was captured
        inside this scope, so the graph's conv launches always see the current weights."""
        if _L().pa2d_get_gemm_mode() != self.mode:
            raise RuntimeError("the GEMM engine (pa2d_set_gemm_mode) changed since these weight packs / this hipGraph "
                               "were made; capture again under the new engine")
        for (_, _, B, H, W, Cc, direction), (wx, wf, pack) in self.packs.items():
            _lib.check(_L().pa2d_conv3x3x2_pack(_p(wx), _p(wf), pack.data_ptr(), pack.numel(), B, H, W, Cc, direction,
                                                _stream()), "conv3x3x2_pack")


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


def _conv_pack(wx, wf, B, H, W, Cc, direction):
    """Pack pointer for the active scope (0 = let the conv call pack into its workspace)."""
    if not _frozen:
        return 0
    scope = _frozen[-1]
    if _L().pa2d_get_gemm_mode() != scope.mode:
        raise RuntimeError("pa2d_set_gemm_mode() was called inside a weights_frozen() scope")
    key = (wx.data_ptr(), wf.data_ptr(), B, H, W, Cc, direction)
    hit = scope.packs.get(key)
    if hit is None:
        nb = _L().pa2d_conv3x3x2_pack_bytes(Cc)
        pack = torch.empty(nb, dtype=torch.uint8, device=wx.device)
        _lib.check(_L().pa2d_conv3x3x2_pack(_p(wx), _p(wf), pack.data_ptr(), nb, B, H, W, Cc, direction, _stream()),
                   "conv3x3x2_pack")
        hit = scope.packs[key] = (wx, wf, pack)
    return hit[2].data_ptr()


def conv3x3x2_fwd(xn, wx, bx, wf, bf, H, W):
    """xn [B,N,C] -> [B,N,2C] = [x_mid | fx_mid]."""
    _chk(xn, wx, bx, wf, bf)
    B, N, Cc = xn.shape
    out = torch.empty(B, N, 2 * Cc, dtype=torch.float32, device=xn.device)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 0)
    nb = _L().pa2d_conv3x3x2_fwd_workspace(B, H, W, Cc)
    ws = _ws(nb, xn)
    e0, e1 = _conv_events()
    _lib.check(_L().pa2d_conv3x3x2_fwd(_p(xn), _p(wx), _p(bx), _p(wf), _p(bf), _p(out), pre, ws.data_ptr(), nb,
                                       B, H, W, Cc, _stream(), e0, e1), "conv3x3x2_fwd")
    return out


def conv3x3x2_bwd(dout, xn, wx, wf, H, W, need_dx=True):
    _chk(dout, xn, wx, wf)
    B, N, Cc = xn.shape
    dxn = torch.empty_like(xn) if need_dx else None
    dwx, dwf = torch.empty_like(wx), torch.empty_like(wf)
    dbx = torch.empty(Cc, dtype=torch.float32, device=xn.device)
    dbf = torch.empty_like(dbx)
    nb = _L().pa2d_conv3x3x2_workspace(B, H, W, Cc)
    ws = _ws(nb, xn)
    pre = _conv_pack(wx, wf, B, H, W, Cc, 1) if need_dx else 0
    e0, e1 = _conv_events() if need_dx else (0, 0)
    _lib.check(_L().pa2d_conv3x3x2_bwd(_p(dout), _p(xn), _p(wx), _p(wf), _p(dxn), _p(dwx), _p(dbx), _p(dwf), _p(dbf),
                                       pre, ws.data_ptr(), nb, B, H, W, Cc, _stream(), e0, e1), "conv3x3x2_bwd")
    return dxn, dwx, dbx, dwf, dbf


def slice_nchunk(B, N, heads):
    return _L().pa2d_slice_nchunk(B, N, heads)


def slice_scatter(xm, ldx, xm_off, v, ldv, v_off, ws_w, bs, temperature, B, N, heads, D, M, want_norm=True,
                  clamp=True):
    """Partial sums of W^T V per point chunk.  xm / v are base tensors; *_off are float offsets."""
    _chk(xm, v, ws_w, bs, temperature)
    nchunk = slice_nchunk(B, N, heads)
    spart = torch.empty(B * heads, nchunk, M, D, dtype=torch.float32, device=xm.device)
    npart = torch.empty(B * heads, nchunk, M, dtype=torch.float32, device=xm.device) if want_norm else None
    _lib.check(_L().pa2d_slice_scatter(_p(xm, xm_off), ldx, _p(v, v_off), ldv, _p(ws_w), _p(bs), _p(temperature),
                                       _p(spart), _p(npart), B, N, heads, D, M, int(clamp), _stream()), "slice_scatter")
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


def token_attn_bwd(s, nrm, wq, wk, wv, dopart):
    _chk(s, nrm, wq, wk, wv, dopart)
    BH, nchunk, M, D = dopart.shape
    ds = torch.empty_like(s)
    dn = torch.empty_like(nrm)
    dwq, dwk, dwv = torch.empty(3, D, D, dtype=torch.float32, device=s.device).unbind(0)   # one block: reduced in place
    nb = _L().pa2d_token_attn_bwd_workspace(BH, D)
    ws = _ws(nb, s)
    _lib.check(_L().pa2d_token_attn_bwd(_p(s), _p(nrm), _p(wq), _p(wk), _p(wv), _p(dopart), _p(ds), _p(dn), _p(dwq),
                                        _p(dwk), _p(dwv), ws.data_ptr(), nb, BH, nchunk, M, D, _stream()),
               "token_attn_bwd")
    return ds, dn, dwq, dwk, dwv


def deslice_fwd(xm, ldx, xm_off, o, ws_w, bs, temperature, B, N, heads, D, M, clamp=True):
    _chk(xm, o, ws_w, bs, temperature)
    y = torch.empty(B, N, heads * D, dtype=torch.float32, device=xm.device)
    _lib.check(_L().pa2d_deslice_fwd(_p(xm, xm_off), ldx, _p(o), _p(ws_w), _p(bs), _p(temperature), _p(y), heads * D,
                                     B, N, heads, D, M, int(clamp), _stream()), "deslice_fwd")
    return y


def slice_bwd_points(xf, dy, ws_w, bs, temperature, o, ds, dn, B, N, heads, D, M, clamp=True):
    """xf = [B,N,2C] ([x_mid | fx_mid]); returns dxf [B,N,2C], dws, dbs, dtemperature [heads]."""
    _chk(xf, dy, ws_w, bs, temperature, o, ds, dn)
    Cc = heads * D
    dxf = torch.empty_like(xf)
    dws = torch.empty_like(ws_w)
    dbs = torch.empty_like(bs)
    dtemp = torch.empty(heads, dtype=torch.float32, device=xf.device)
    nb = _L().pa2d_slice_bwd_workspace(B, N, heads, D, M)
    ws = _ws(nb, xf)
    _lib.check(_L().pa2d_slice_bwd_points(_p(xf), 2 * Cc, _p(xf, Cc), 2 * Cc, _p(dy), Cc, _p(ws_w), _p(bs),
                                          _p(temperature), _p(o), _p(ds), _p(dn), _p(dxf), 2 * Cc, _p(dxf, Cc),
                                          2 * Cc, _p(dws), _p(dbs), _p(dtemp), ws.data_ptr(), nb, B, N, heads, D, M,
                                          int(clamp), _stream()), "slice_bwd_points")
    return dxf, dws, dbs, dtemp


def head_fwd(xn2d, w, b):
    _chk(xn2d, w, b)
    rows, Cc = xn2d.shape
    O = w.shape[0]
    y = torch.empty(rows, O, dtype=torch.float32, device=xn2d.device)
    _lib.check(_L().pa2d_head_fwd(_p(xn2d), _p(w), _p(b), _p(y), rows, Cc, O, _stream()), "head_fwd")
    return y


def head_bwd(dy, xn2d, w):
    _chk(dy, xn2d, w)
    rows, Cc = xn2d.shape
    O = w.shape[0]
    dxn = torch.empty_like(xn2d)
    dw = torch.empty_like(w)
    db = torch.empty(O, dtype=torch.float32, device=w.device)
    nb = _L().pa2d_head_bwd_workspace(rows, Cc, O)
    ws = _ws(nb, dy)
    _lib.check(_L().pa2d_head_bwd(_p(dy), _p(xn2d), _p(w), _p(dxn), _p(dw), _p(db), ws.data_ptr(), nb, rows, Cc, O,
                                  _stream()), "head_bwd")
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
    _chk(pred2d, y2d, dn, yn, gout)
    B, L = pred2d.shape
    dpred = torch.empty_like(pred2d)
    _lib.check(_L().pa2d_rel_l2_bwd(_p(pred2d), _p(y2d), _p(dn), _p(yn), _p(gout), _p(dpred), B, L, _stream()),
               "rel_l2_bwd")
    return dpred
