"""Stage-by-stage parity of the HIP kernels (called through the libpa2d C ABI via ops.py) against
the CPU oracle on identical seeded inputs.  fp32 tolerances: the oracle is evaluated in fp64, the
kernels in fp32 -> rel-L2 <= 2e-6 for forward stages, <= 1e-5 for gradients (SURVEY §8c calibrates
the reference's own fp32-vs-fp64 error at 1e-6 / 3e-7..2.5e-4)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

FWD_TOL = 3e-6
BWD_TOL = 2e-5


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


def _r(rng, *shape, scale=1.0):
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


@pytest.mark.parametrize("rows,C", [(7, 32), (300, 64), (4096, 256), (1000, 1024)])
def test_layernorm(dev, rows, C):
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    rng = np.random.default_rng(rows + C)
    x, g, b, dy, dres = _r(rng, rows, C) * 2 + 0.5, 1 + 0.1 * _r(rng, C), 0.1 * _r(rng, C), _r(rng, rows, C), _r(rng, rows, C)
    xd = x.double().requires_grad_(True)
    gd, bd = g.double().requires_grad_(True), b.double().requires_grad_(True)
    yo = orc.layer_norm(xd, gd, bd)
    yo.backward(dy.double())
    y, mean, rstd = ops.layernorm_fwd(x.to(dev), g.to(dev), b.to(dev))
    assert rel_l2(y, yo) < FWD_TOL
    dx, dg, db = ops.layernorm_bwd(dy.to(dev), x.to(dev), mean, rstd, g.to(dev), dres.to(dev))
    assert rel_l2(dx, xd.grad + dres.double()) < BWD_TOL
    assert rel_l2(dg, gd.grad) < BWD_TOL and rel_l2(db, bd.grad) < BWD_TOL


ENGINES = ["f32", "split"]        # both fp32-accurate engines must meet the SAME tolerances


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("M,N,K,act", [(60, 32, 12, "gelu"), (257, 64, 76, "silu"), (4096, 256, 256, "gelu"),
                                       (1000, 512, 76, None), (333, 96, 200, "tanh"), (130, 8, 64, None)])
def test_linear(dev, M, N, K, act, engine):
    _check_linear(dev, M, N, K, act, engine, FWD_TOL, BWD_TOL)


# large-M linears of the split engine run on the persistent row-panel kernel (>= 256 tiles of 256 x 128): residual
# preloaded into the accumulators, GELU / GELU' fast epilogues, ragged last row block and ragged columns, generic epilogue
@pytest.mark.parametrize("M,N,K,act", [(70000, 256, 256, "gelu"), (65536, 256, 256, None), (66001, 192, 128, None),
                                       (65553, 128, 512, "silu"), (65539, 200, 96, None), (65539, 200, 96, "gelu")])
def test_linear_row_panel(dev, M, N, K, act):
    _check_linear(dev, M, N, K, act, "split", FWD_TOL, BWD_TOL)


# Row-stationary kernel of the split engine (K in {128, 256}, N % 128 == 0, >= 256 rounds of 128 rows, scratch from
# pa2d_gemm_fwd_workspace / pa2d_gemm_bwd_data_workspace): every epilogue (plain, bias, bias + residual, GELU with and
# without the saved pre-activation, GELU' data gradient), both K, 2 / 4 / 8 column pairs, one and several rounds per
# workgroup, a ragged tail of < 128 rows; and the same calls with the kernel switched off agree to fp32 rounding.
@pytest.mark.parametrize("M,N,K", [(32768, 128, 128), (33285, 256, 128), (40000, 128, 256), (65536 + 128 * 5 + 77, 256, 256),
                                   (33000, 512, 256), (32768 + 130, 1024, 128)])
def test_linear_row_stationary(dev, kernel_env, M, N, K):
    from transformerbasednavierstokesolver_amd import ops, _lib
    from oracle import transolver_oracle as orc
    assert _lib.load().pa2d_gemm_fwd_workspace(N, K, 1) == N * K * 6
    rng = np.random.default_rng(M + N + K)
    x, w, b, res = _r(rng, M, K), _r(rng, N, K, scale=K ** -0.5), 0.1 * _r(rng, N), _r(rng, M, N)
    dy, pre2 = _r(rng, M, N), _r(rng, M, K)
    xd, wd, bd = x.to(dev).double(), w.to(dev).double(), b.to(dev).double()
    lin = xd @ wd.t()
    p2 = pre2.to(dev).double().requires_grad_(True)
    orc._ACTS["gelu"](p2).backward(torch.ones(M, K, dtype=torch.float64, device=dev))
    want = {"plain": lin, "bias": lin + bd, "bias_res": lin + bd + res.to(dev).double(),
            "gelu": orc._ACTS["gelu"](lin + bd), "pre": lin + bd,
            "bwd": dy.to(dev).double() @ wd, "bwd_gelu": (dy.to(dev).double() @ wd) * p2.grad}

    X, W, Bv, R, DY, P2 = (t.to(dev) for t in (x, w, b, res, dy, pre2))

    def run():
        out = {"plain": ops.linear_fwd(X, W, engine="split")[0], "bias": ops.linear_fwd(X, W, Bv, engine="split")[0],
               "bias_res": ops.linear_fwd(X, W, Bv, res=R, engine="split")[0]}
        out["gelu"], out["pre"] = ops.linear_fwd(X, W, Bv, act="gelu", want_pre=True, engine="split")
        out["gelu_nopre"] = ops.linear_fwd(X, W, Bv, act="gelu", engine="split")[0]
        out["bwd"] = ops.linear_bwd_data(DY, W, engine="split")      # dx[M, K] = dy[M, N] . w[N, K]
        out["bwd_gelu"] = ops.linear_bwd_data(DY, W, pre=P2, act="gelu", engine="split")
        return out

    got = run()
    for k in ("plain", "bias", "bias_res", "gelu", "pre"):
        assert rel_l2(got[k], want[k]) < FWD_TOL, k
    assert torch.equal(got["gelu_nopre"], got["gelu"])
    assert rel_l2(got["bwd"], want["bwd"]) < BWD_TOL and rel_l2(got["bwd_gelu"], want["bwd_gelu"]) < BWD_TOL
    with ops.weights_frozen() as scope:      # images made once per weight (pa2d_gemm_weight_image) and passed ready-made
        frozen = run()
        frozen2 = run()
        assert len(scope.images) == (2 if K <= 256 and N <= 256 else 1)      # forward image, transposed image
    for k in got:
        assert torch.equal(frozen[k], got[k]) and torch.equal(frozen2[k], got[k]), k
    kernel_env(PA2D_LIN_ROWPANEL="off")
    ref = run()
    for k in got:
        assert rel_l2(got[k], ref[k]) < 1e-6, k


@pytest.mark.parametrize("engine", ENGINES + ["bf16s"])
@pytest.mark.parametrize("M,N,K,act", [(40000, 256, 256, "gelu"), (4096, 256, 256, "gelu"), (333, 96, 200, "silu"),
                                       (65536, 128, 128, "gelu")])
def test_linear_saved_derivative(dev, M, N, K, act, engine):
    """PA2D_ACT_SAVE_DERIVATIVE: the forward saves act'(pre-activation) instead of the pre-activation and the data gradient
    multiplies by it — every kernel family (row-stationary, panel, per-tile split and exact, bf16 storage) against fp64
    and against the plain pre-activation form."""
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    bf = engine == "bf16s"
    if bf and (K % 32 or N % 32):
        pytest.skip("bf16 storage needs widths that are multiples of 32")
    tol_f, tol_b = (2e-2, 2e-2) if bf else (FWD_TOL, BWD_TOL)
    cast = (lambda t: t.to(dev).bfloat16()) if bf else (lambda t: t.to(dev))
    eng = None if bf else engine
    rng = np.random.default_rng(M + N)
    x, w, b, dy = _r(rng, M, K), _r(rng, N, K, scale=K ** -0.5), 0.1 * _r(rng, N), _r(rng, M, N)
    pre = (cast(x).double() @ w.to(dev).double().t() + b.to(dev).double()).requires_grad_(True)
    orc._ACTS[act](pre).backward(torch.ones_like(pre))
    y, d = ops.linear_fwd(cast(x), w.to(dev), b.to(dev), act=act, want_pre=True, engine=eng, save_derivative=True)
    y0, p0 = ops.linear_fwd(cast(x), w.to(dev), b.to(dev), act=act, want_pre=True, engine=eng)
    assert torch.equal(y, y0)
    assert rel_l2(d, pre.grad) < (3e-2 if bf else 1e-5)
    w2 = _r(rng, N, K, scale=N ** -0.5).to(dev)      # dx[M, K2] = (dy2 . w2) * act'(pre2), pre2 [M, K2]: reuse d as [M, N] with K2 = N
    dy2 = cast(_r(rng, M, K))
    got = ops.linear_bwd_data(dy2, w2.t().contiguous(), pre=d, act=act, engine=eng, pre_is_derivative=True)
    ref = ops.linear_bwd_data(dy2, w2.t().contiguous(), pre=p0, act=act, engine=eng)
    want = (dy2.double() @ w2.t().double()) * pre.grad
    assert rel_l2(got, want) < tol_b and rel_l2(got, ref) < (3e-2 if bf else 1e-5)


def test_linear_row_stationary_padded_leading_dimensions(dev):
    """The C ABI takes leading dimensions: x, y, res and pre as column windows of wider buffers (ld = width + pad, a
    multiple of 4 floats as the 16-byte epilogue needs; other pads fall back to the per-tile kernels) — same results,
    and nothing outside the windows is written."""
    from transformerbasednavierstokesolver_amd import ops, _lib
    L = _lib.load()
    M, N, K = 32768 + 64, 256, 256
    rng = np.random.default_rng(5)
    w, b = _r(rng, N, K, scale=K ** -0.5).to(dev), (0.1 * _r(rng, N)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    nb = L.pa2d_gemm_fwd_workspace(N, K, 1)
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    for pad in (8, 4, 6):                     # 6: the output windows are not 16-byte aligned -> per-tile kernels
        xpad = 8
        xb, rb = _r(rng, M, K + xpad).to(dev), _r(rng, M, N + pad).to(dev)
        yb = torch.full((M, N + pad), 7.0, device=dev)
        pb = torch.full((M, N + pad), 9.0, device=dev)
        want = xb[:, :K].double() @ w.double().t() + b.double()
        # plain + residual
        rc = L.pa2d_gemm_bias_act_fwd(xb.data_ptr(), K + xpad, w.data_ptr(), K, b.data_ptr(), rb.data_ptr(), N + pad, yb.data_ptr(),
                                      N + pad, 0, 0, 0, ws.data_ptr(), nb, M, N, K, 0, 1, st)
        assert rc == 0
        assert rel_l2(yb[:, :N], want + rb[:, :N].double()) < FWD_TOL and bool((yb[:, N:] == 7.0).all())
        # GELU with the saved pre-activation
        yb.fill_(7.0)
        rc = L.pa2d_gemm_bias_act_fwd(xb.data_ptr(), K + xpad, w.data_ptr(), K, b.data_ptr(), 0, 0, yb.data_ptr(), N + pad,
                                      pb.data_ptr(), N + pad, 0, ws.data_ptr(), nb, M, N, K, ops.ACT_IDS["gelu"], 1, st)
        assert rc == 0
        assert rel_l2(pb[:, :N], want) < FWD_TOL and bool((pb[:, N:] == 9.0).all()) and bool((yb[:, N:] == 7.0).all())
        assert rel_l2(yb[:, :N], torch.nn.functional.gelu(want)) < FWD_TOL


def _check_linear(dev, M, N, K, act, engine, FWD_TOL, BWD_TOL):
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    rng = np.random.default_rng(M + N + K)
    x, w, b, res = _r(rng, M, K), _r(rng, N, K, scale=K ** -0.5), 0.1 * _r(rng, N), _r(rng, M, N)
    dy = _r(rng, M, N)
    xd, wd, bd = (t.double().requires_grad_(True) for t in (x, w, b))
    pre = xd @ wd.t() + bd
    yo = (orc._ACTS[act](pre) if act else pre) + res.double()
    yo.backward(dy.double())
    y, pre_k = ops.linear_fwd(x.to(dev), w.to(dev), b.to(dev), res=res.to(dev), act=act, want_pre=True, engine=engine)
    assert rel_l2(y, yo) < FWD_TOL
    assert rel_l2(pre_k, pre) < FWD_TOL
    # backward pieces (the act' factor is applied by the NEXT bwd_data call's epilogue, tested below)
    dwk, dbk = ops.linear_bwd_weight(dy.to(dev), x.to(dev), engine=engine)
    dxk = ops.linear_bwd_data(dy.to(dev), w.to(dev), engine=engine)
    # accumulate = 1: the same call ADDS to a caller buffer (the flat gradient bucket)
    acc_w, acc_b = torch.full_like(dwk, 0.5), torch.full_like(dbk, -0.25)
    ops.linear_bwd_weight(dy.to(dev), x.to(dev), engine=engine, into=(acc_w, acc_b))
    assert rel_l2(acc_w - 0.5, dwk) < 1e-5 and rel_l2(acc_b + 0.25, dbk) < 1e-5
    assert rel_l2(dbk, dy.double().sum(0)) < BWD_TOL          # the bias gradient rides in the weight-gradient GEMM
    if act is None:
        assert rel_l2(dwk, wd.grad) < BWD_TOL and rel_l2(dbk, bd.grad) < BWD_TOL
        assert rel_l2(dxk, xd.grad) < BWD_TOL
    else:
        # check the fused dact epilogue: dx = (dy . w) * act'(pre2) with an independent pre2 [M,K]
        pre2 = _r(rng, M, K)
        p2 = pre2.double().requires_grad_(True)
        orc._ACTS[act](p2).backward(torch.ones(M, K, dtype=torch.float64))
        want = (dy.double() @ w.double()) * p2.grad
        got = ops.linear_bwd_data(dy.to(dev), w.to(dev), pre=pre2.to(dev), act=act, engine=engine)
        assert rel_l2(got, want) < BWD_TOL
        assert rel_l2(dwk, dy.double().t() @ x.double()) < BWD_TOL


# C = 256 / 128: pre-split planes + transposed-read weight gradient on the split engine; C = 64 / 192: planes for the
# data GEMMs, register-split gather fallback for the weight gradient; C = 32: small tiles stay on the exact kernels
@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("B,H,W,C", [(2, 6, 5, 32), (1, 64, 64, 64), (2, 64, 64, 256), (1, 21, 17, 128), (2, 16, 16, 64),
                                     (1, 12, 20, 192)])
def test_conv3x3x2(dev, B, H, W, C, engine):
    _check_conv(dev, B, H, W, C, engine, FWD_TOL, BWD_TOL)


def _check_conv(dev, B, H, W, C, engine, FWD_TOL, BWD_TOL):
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    rng = np.random.default_rng(B * H + C)
    N = H * W
    xn = _r(rng, B, N, C)
    wx, wf = _r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5), _r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5)
    bx, bf = 0.1 * _r(rng, C), 0.1 * _r(rng, C)
    dout = _r(rng, B, N, 2 * C)
    xd, wxd, wfd, bxd, bfd = (t.double().requires_grad_(True) for t in (xn, wx, wf, bx, bf))
    out_o = torch.cat([orc.conv3x3(xd, wxd, bxd, H, W), orc.conv3x3(xd, wfd, bfd, H, W)], -1)
    out_o.backward(dout.double())
    out = ops.conv3x3x2_fwd(xn.to(dev), wx.to(dev), bx.to(dev), wf.to(dev), bf.to(dev), H, W, engine=engine)
    assert rel_l2(out, out_o) < FWD_TOL
    dxn, dwx, dbx, dwf, dbf = ops.conv3x3x2_bwd(dout.to(dev), xn.to(dev), wx.to(dev), wf.to(dev), H, W, engine=engine)
    into = tuple(torch.full_like(t, 0.125) for t in (dwx, dbx, dwf, dbf))      # accumulate = 1 adds to the buffers
    ops.conv3x3x2_bwd(dout.to(dev), xn.to(dev), wx.to(dev), wf.to(dev), H, W, need_dx=False, engine=engine, into=into)
    for got, ref in zip(into, (dwx, dbx, dwf, dbf)):
        assert rel_l2(got - 0.125, ref) < 1e-5
    assert rel_l2(dxn, xd.grad) < BWD_TOL
    assert rel_l2(dwx, wxd.grad) < BWD_TOL and rel_l2(dwf, wfd.grad) < BWD_TOL
    assert rel_l2(dbx, bxd.grad) < BWD_TOL and rel_l2(dbf, bfd.grad) < BWD_TOL


SLICE_CASES = [  # B, N, heads, D, M
    (2, 30, 4, 8, 12),      # tiny, ragged points (30 % 16 != 0), M not a multiple of 16
    (1, 4096, 8, 8, 32),    # shipped-checkpoint geometry (C=64)
    (2, 4096, 8, 32, 64),   # NS benchmark geometry (C=256)
    (1, 1000, 8, 16, 128),  # Darcy-like geometry (D=16, M=128), ragged N
    (1, 200, 2, 64, 20),
]


def _slice_inputs(B, N, heads, D, M, seed):
    rng = np.random.default_rng(seed)
    C = heads * D
    xf = _r(rng, B, N, 2 * C)
    ws, bs = _r(rng, M, D, scale=D ** -0.5), 0.3 * _r(rng, M)
    temp = torch.tensor(np.resize(np.array([0.03, 0.5, 7.0, 0.25, 1.5, 0.1, 5.0, 0.8], dtype=np.float32), heads))
    wq, wk, wv = (_r(rng, D, D, scale=1.5 * D ** -0.5) for _ in range(3))
    dy = _r(rng, B, N, C)
    return xf, ws, bs, temp, wq, wk, wv, dy


@pytest.mark.parametrize("engine", ENGINES)       # f32: exact v_mfma_f32_16x16x4_f32 kernels; split: bf16 MFMA on exact 3-plane splits
@pytest.mark.parametrize("B,N,heads,D,M", SLICE_CASES)
def test_slice_token_deslice_forward(dev, B, N, heads, D, M, engine):
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    C = heads * D
    xf, ws, bs, temp, wq, wk, wv, _ = _slice_inputs(B, N, heads, D, M, B + N + M)
    xfd = xf.double()
    w, norm, s, tok = orc.slice_tokens(xfd[..., :C], xfd[..., C:], ws.double(), bs.double(), temp.double(), heads)
    o = orc.token_attention(tok, wq.double(), wk.double(), wv.double())
    y = orc.deslice(w, o)
    g = lambda t: t.to(dev)
    spart, npart = ops.slice_scatter(g(xf), 2 * C, 0, g(xf), 2 * C, C, g(ws), g(bs), g(temp), B, N, heads, D, M, engine=engine)
    assert rel_l2(spart.sum(1).view(B, heads, M, D), s) < FWD_TOL
    assert rel_l2(npart.sum(1).view(B, heads, M), norm) < FWD_TOL
    sk, nk, ok = ops.token_attn_fwd(spart, npart, g(wq), g(wk), g(wv))
    assert rel_l2(sk.view(B, heads, M, D), s) < FWD_TOL and rel_l2(nk.view(B, heads, M), norm) < FWD_TOL
    assert rel_l2(ok.view(B, heads, M, D), o) < 2e-5
    # de-slice checked with the ORACLE's tokens so that errors do not compound
    yk = ops.deslice_fwd(g(xf), 2 * C, 0, g(o.float().reshape(B * heads, M, D).contiguous()), g(ws), g(bs), g(temp),
                         B, N, heads, D, M, engine=engine)
    assert rel_l2(yk, y) < FWD_TOL


@pytest.mark.parametrize("engine", ENGINES)
@pytest.mark.parametrize("B,N,heads,D,M", SLICE_CASES)
def test_slice_core_backward(dev, B, N, heads, D, M, engine):
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    C = heads * D
    xf, ws, bs, temp, wq, wk, wv, dy = _slice_inputs(B, N, heads, D, M, 7 * B + N + M)
    d = lambda t: t.double()
    ref = orc.slice_core_backward(d(xf[..., :C]), d(xf[..., C:]), d(dy), d(ws), d(bs), d(temp), d(wq), d(wk), d(wv), heads)
    g = lambda t: t.to(dev).contiguous()
    # forward state
    spart, npart = ops.slice_scatter(g(xf), 2 * C, 0, g(xf), 2 * C, C, g(ws), g(bs), g(temp), B, N, heads, D, M, engine=engine)
    s, nrm, o = ops.token_attn_fwd(spart, npart, g(wq), g(wk), g(wv))
    # phase A: dO partials
    dopart, _ = ops.slice_scatter(g(xf), 2 * C, 0, g(dy), C, 0, g(ws), g(bs), g(temp), B, N, heads, D, M, want_norm=False,
                                  engine=engine)
    assert rel_l2(dopart.sum(1).view(B, heads, M, D), ref["do"]) < BWD_TOL
    ds, dn, dwq, dwk, dwv = ops.token_attn_bwd(s, nrm, g(wq), g(wk), g(wv), dopart)
    assert rel_l2(ds.view(B, heads, M, D), ref["ds"]) < 5e-5
    assert rel_l2(dn.view(B, heads, M), ref["dn"]) < 5e-5
    assert rel_l2(dwq, ref["dwq"]) < 5e-5 and rel_l2(dwk, ref["dwk"]) < 5e-5 and rel_l2(dwv, ref["dwv"]) < 5e-5
    # phase C with the oracle's dS / dn / O so that errors do not compound
    f32 = lambda t, *shape: g(t.float().reshape(*shape))
    o_ref = orc.token_attention(orc.slice_tokens(d(xf[..., :C]), d(xf[..., C:]), d(ws), d(bs), d(temp), heads)[3],
                                d(wq), d(wk), d(wv))
    dxf, dws, dbs, dtemp = ops.slice_bwd_points(g(xf), g(dy), g(ws), g(bs), g(temp), f32(o_ref, B * heads, M, D),
                                                f32(ref["ds"], B * heads, M, D), f32(ref["dn"], B * heads, M),
                                                B, N, heads, D, M, engine=engine)
    assert rel_l2(dxf[..., :C], ref["dxm"]) < BWD_TOL
    assert rel_l2(dxf[..., C:], ref["dfm"]) < BWD_TOL
    assert rel_l2(dws, ref["dws"]) < BWD_TOL and rel_l2(dbs, ref["dbs"]) < BWD_TOL
    assert rel_l2(dtemp, ref["dtemperature"].reshape(heads)) < BWD_TOL
    # clamp mask: heads whose raw temperature is outside [0.1, 5] get exactly zero gradient
    outside = (temp < 0.1) | (temp > 5.0)
    assert torch.all(dtemp.cpu()[outside] == 0)


@pytest.mark.parametrize("rows,C,O", [(30, 32, 2), (4096, 256, 1), (500, 64, 5)])
def test_head(dev, rows, C, O):
    from transformerbasednavierstokesolver_amd import ops
    rng = np.random.default_rng(rows + O)
    x, w, b, dy = _r(rng, rows, C), _r(rng, O, C, scale=C ** -0.5), 0.1 * _r(rng, O), _r(rng, rows, O)
    g = lambda t: t.to(dev)
    y = ops.head_fwd(g(x), g(w), g(b))
    assert rel_l2(y, x.double() @ w.double().t() + b.double()) < FWD_TOL
    dx, dw, db = ops.head_bwd(g(dy), g(x), g(w))
    assert rel_l2(dx, dy.double() @ w.double()) < BWD_TOL
    assert rel_l2(dw, dy.double().t() @ x.double()) < BWD_TOL
    assert rel_l2(db, dy.double().sum(0)) < BWD_TOL


@pytest.mark.parametrize("policy", ["force", "off"])
@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 256), (1, 21, 17, 128), (2, 16, 16, 64), (1, 12, 20, 192), (3, 8, 32, 32),
                                     (1, 45, 70, 64), (2, 24, 24, 192), (1, 40, 30, 128)])
def test_conv_split_engine_both_kernels(dev, kernel_env, B, H, W, C, policy):
    """The split engine has two conv kernels: the halo-tile-in-LDS kernel (spatial 8x32 tiles; picked when the launch
    fills the chip) and the plain implicit-GEMM kernel.  PA2D_CONV_HALO forces either one on shapes with ragged
    tiles in both directions, several images and 1..8 channel chunks: same fp32 tolerances.  Likewise the weight
    gradient: 256x256-tile planes kernel (default when 2C >= 256; the last two shapes have ragged row AND column
    tiles and column groups whose taps change inside a tile) vs the 128x128 one (PA2D_MC_BIG=off)."""
    kernel_env(PA2D_CONV_HALO=policy, PA2D_MC_BIG="off" if policy == "off" else None)
    _check_conv(dev, B, H, W, C, "split", FWD_TOL, BWD_TOL)


@pytest.mark.parametrize("B,H,W,C", [(2, 64, 64, 256), (1, 21, 17, 128), (1, 45, 70, 64)])
def test_conv_halo_both_mfma_shapes(dev, kernel_env, B, H, W, C):
    """The halo conv's consumers exist on v_mfma_f32_16x16x32_bf16 (default: faster on real data) and on 32x32x16
    (PA2D_CONV_MFMA=32): both meet the fp32 tolerances, also on ragged tiles, and agree to fp32 rounding."""
    from transformerbasednavierstokesolver_amd import ops
    rng = np.random.default_rng(B + H + C)
    xn, wx, wf = _r(rng, B, H * W, C).to(dev), (_r(rng, C, C, 3, 3) * 0.05).to(dev), (_r(rng, C, C, 3, 3) * 0.05).to(dev)
    bx, bf = _r(rng, C).to(dev), _r(rng, C).to(dev)
    outs = {}
    for shape in ("16", "32"):
        kernel_env(PA2D_CONV_HALO="force", PA2D_CONV_MFMA=shape)
        _check_conv(dev, B, H, W, C, "split", FWD_TOL, BWD_TOL)
        outs[shape] = ops.conv3x3x2_fwd(xn, wx, bx, wf, bf, H, W, engine="split")
    assert rel_l2(outs["16"], outs["32"]) < 1e-6


def test_bf16_compute_engine_stage_tolerances(dev, kernel_env):
    kernel_env(PA2D_CONV_HALO="force")
    _check_conv(dev, 2, 64, 64, 256, "bf16", 1e-2, 1e-2)
    _check_conv(dev, 1, 21, 17, 128, "bf16", 1e-2, 1e-2)
    kernel_env(PA2D_CONV_HALO=None)
    _bf16_stage_cases(dev)


def _bf16_stage_cases(dev):
    """engine "bf16": operands rounded to bf16, one bf16 MFMA term, fp32 accumulate/storage.
    SURVEY 8c: bf16 forward tolerance 3e-2 (the reference under bf16 autocast is 1.4-1.6e-2 from fp64);
    a single GEMM stage stays below 1e-2."""
    for args in ((2, 64, 64, 256), (1, 21, 17, 128), (2, 16, 16, 64), (1, 12, 20, 192)):
        _check_conv(dev, *args, "bf16", 1e-2, 1e-2)
    _check_linear(dev, 4096, 256, 256, "gelu", "bf16", 1e-2, 1e-2)
    _check_linear(dev, 1000, 512, 76, None, "bf16", 1e-2, 1e-2)
    _check_linear(dev, 66001, 256, 256, "gelu", "bf16", 1e-2, 1e-2)        # row-panel kernel, one bf16 term, ragged tail


def test_small_parameter_gradients_accumulate_in_place(dev):
    """`into=`: LayerNorm / head / token / slice parameter gradients are ADDED to caller buffers by the reduce pass."""
    from transformerbasednavierstokesolver_amd import ops
    rng = np.random.default_rng(5)
    rows, C = 300, 64
    x, g, dy = _r(rng, rows, C).to(dev), (1 + 0.1 * _r(rng, C)).to(dev), _r(rng, rows, C).to(dev)
    _, mean, rstd = ops.layernorm_fwd(x, g, torch.zeros_like(g))
    _, dg, db = ops.layernorm_bwd(dy, x, mean, rstd, g)
    bufs = (torch.full_like(dg, 2.0), torch.full_like(db, -1.0))
    ops.layernorm_bwd(dy, x, mean, rstd, g, into=bufs)
    assert rel_l2(bufs[0] - 2.0, dg) < 1e-5 and rel_l2(bufs[1] + 1.0, db) < 1e-5
    w, d1 = _r(rng, 2, C).to(dev), _r(rng, rows, 2).to(dev)
    _, dw, dbh = ops.head_bwd(d1, x, w)
    bufs = (torch.full_like(dw, 0.5), torch.full_like(dbh, 0.5))
    ops.head_bwd(d1, x, w, into=bufs)
    assert rel_l2(bufs[0] - 0.5, dw) < 1e-5 and rel_l2(bufs[1] - 0.5, dbh) < 1e-5


def test_split_engine_adversarial_operands(dev):
    """VERDICT r1 ruling, condition (ii): the fp32-accurate split engine against the exact-fp32-MFMA engine on
    adversarial conv operands, both measured against fp64.
      * per-pixel dynamic range 2^+-40 (neighbouring pixels of wildly different magnitude meet in one 3x3 window),
      * every value scaled to 2^-90 (far below any activation or gradient this model produces),
      * exact zeros (whole pixels and whole channels),
      * +-inf / NaN propagation.
    Asserted: split error <= 2 x exact error (+ 1e-7: both sit at the fp32 rounding floor), zeros give exactly the
    bias on both, and the non-finite outputs coincide.  DOCUMENTED LIMIT (not rejected: it would take a reduction
    pass over every operand to detect): an operand whose magnitude is below 2^-102 loses its `lo`, then its `mid`
    plane to bf16 underflow, so the split degrades gracefully from 24 to 8 significant bits between 2^-102 and 2^-126
    — asserted here as 'finite and <= 2^-7 relative at 2^-120'.  LayerNorm outputs are O(1) and gradients of a rel-L2
    loss O(1e-8..1), so the path never gets near; data that does should use engine "f32"."""
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    B, H, W, C = 2, 24, 24, 128
    N = H * W
    rng = np.random.default_rng(99)
    wx, wf = _r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5), _r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5)
    bx, bf = 0.1 * _r(rng, C), 0.1 * _r(rng, C)
    g = lambda t: t.to(dev)

    zb = torch.zeros(C)

    def run(xn, engine, bias=True):
        b1, b2 = (bx, bf) if bias else (zb, zb)
        return ops.conv3x3x2_fwd(g(xn), g(wx), g(b1), g(wf), g(b2), H, W, engine=engine)

    def ref(xn, bias=True):
        xd = xn.double()
        b1, b2 = (bx, bf) if bias else (zb, zb)
        return torch.cat([orc.conv3x3(xd, wx.double(), b1.double(), H, W), orc.conv3x3(xd, wf.double(), b2.double(), H, W)], -1)

    base = _r(rng, B, N, C)
    # (1) dynamic range 2^+-40 per pixel
    scale = torch.from_numpy(np.exp2(rng.uniform(-40, 40, size=(B, N, 1))).astype(np.float32))
    for name, xn, with_bias in (("range 2^+-40", base * scale, True), ("all at 2^-90", base * 2.0 ** -90, False)):
        r = ref(xn, with_bias)              # (tiny operands: zero bias, or the O(0.1) bias would swamp the products)
        errs = {}
        for e in ("f32", "split"):
            out = run(xn, e, with_bias).double().cpu()
            errs[e] = float(((out - r) * 2.0 ** 80).norm() / (r * 2.0 ** 80).norm())
        assert errs["split"] <= 2 * errs["f32"] + 1e-7, (name, errs)
        assert errs["f32"] < 1e-5, (name, errs)
    # (2) graceful degradation below 2^-102 (documented, see docstring)
    xn = base * 2.0 ** -120
    r = ref(xn, False) * 2.0 ** 110
    out = run(xn, "split", False).double().cpu() * 2.0 ** 110
    assert torch.isfinite(out).all() and float((out - r).norm() / r.norm()) < 2.0 ** -7
    # (3) exact zeros: zero pixels (rows) and zero channels -> bias exactly where the whole window is zero
    xz = base.clone()
    xz[:, : 5 * W] = 0.0                       # first five image rows of every sample
    xz[..., 7] = 0.0
    for e in ("f32", "split"):
        out = run(xz, e).cpu()
        interior = out.reshape(B, H, W, 2 * C)[:, :4]                     # windows entirely inside the zero rows
        assert torch.equal(interior, torch.cat([bx, bf]).expand_as(interior)), e
    # (4) inf / NaN propagation: the non-finite outputs coincide (= the 3x3 neighbourhoods of the poisoned pixels)
    xp = base.clone()
    xp[0, 5 * W + 7, 3] = float("inf")
    xp[1, 11 * W + 2, 0] = float("nan")
    xp[1, 20 * W + 20, 9] = -float("inf")
    masks = {e: torch.isfinite(run(xp, e)).cpu() for e in ("f32", "split")}
    assert torch.equal(masks["f32"], masks["split"])
    bad = (~masks["split"]).reshape(B, H, W, 2 * C).any(-1)
    assert int(bad[0].sum()) == 9 and int(bad[1].sum()) == 18            # exactly the three 3x3 windows


def test_split_engine_adversarial_operands_linear(dev):
    """The same conditions on the large-M plain GEMMs, which the split engine also serves (row-panel forward / data
    gradient kernel, transposed-read weight-gradient kernel): per-row dynamic range 2^+-40, everything at 2^-90, zero rows,
    +-inf / NaN rows.  Asserted: split error <= 2 x exact error + 1e-7 against fp64, zero rows give exactly the bias, the
    non-finite outputs coincide (whole rows of the poisoned elements, nothing else)."""
    from transformerbasednavierstokesolver_amd import ops
    M, N, K = 65536, 256, 256
    rng = np.random.default_rng(7)
    w, b = _r(rng, N, K, scale=K ** -0.5), 0.1 * _r(rng, N)
    base, dy = _r(rng, M, K), _r(rng, M, N)
    g = lambda t: t.to(dev)
    scale = torch.from_numpy(np.exp2(rng.uniform(-40, 40, size=(M, 1))).astype(np.float32))
    zb = torch.zeros(N)
    for name, x, bias in (("range 2^+-40", base * scale, b), ("all at 2^-90", base * 2.0 ** -90, zb)):
        r = x.double() @ w.double().t() + bias.double()
        rw = dy.double().t() @ x.double()
        errs, errw = {}, {}
        for e in ("f32", "split"):
            out = ops.linear_fwd(g(x), g(w), g(bias), engine=e)[0].double().cpu()
            errs[e] = float(((out - r) * 2.0 ** 80).norm() / (r * 2.0 ** 80).norm())
            dw, _ = ops.linear_bwd_weight(g(dy), g(x), engine=e)
            errw[e] = float(((dw.double().cpu() - rw) * 2.0 ** 60).norm() / (rw * 2.0 ** 60).norm())
        assert errs["split"] <= 2 * errs["f32"] + 1e-7 and errs["f32"] < 1e-5, (name, errs)
        assert errw["split"] <= 2 * errw["f32"] + 1e-7 and errw["f32"] < 1e-5, (name, errw)
    xz = base.clone()
    xz[1000:3000] = 0.0
    for e in ("f32", "split"):
        out = ops.linear_fwd(g(xz), g(w), g(b), engine=e)[0].cpu()
        assert torch.equal(out[1000:3000], b.expand(2000, N)), e
    xp = base.clone()
    xp[5, 3] = float("inf")
    xp[40000, 0] = float("nan")
    xp[65535, 255] = -float("inf")
    masks = {e: torch.isfinite(ops.linear_fwd(g(xp), g(w), g(b), engine=e)[0]).cpu() for e in ("f32", "split")}
    assert torch.equal(masks["f32"], masks["split"])
    bad = (~masks["split"]).any(-1)
    assert bad.nonzero().flatten().tolist() == [5, 40000, 65535] and bool((~masks["split"])[5].all())


def test_split_engine_adversarial_operands_slice(dev):
    """The slice stages on the bf16-split kernels (engine "split": logits, scatter, gather and every backward contraction as
    six bf16 MFMA terms of exact 3-plane splits) against the exact-fp32-MFMA kernels (engine "f32") on operands that never
    passed a LayerNorm: x_mid / fx_mid are conv outputs and the logits are divided by a temperature clamped down to 0.1
    (Physics_Attention.py:98-99).  Conditions: per-point dynamic range 2^+-40, raw temperatures at and beyond both clamp
    ends, logits whose spread (+-300) overflows exp without the max subtraction, exact zeros, +-inf / NaN rows.  Asserted
    against fp64: split error <= 2 x exact error (+ the fp32 rounding floor), identical non-finite masks."""
    from transformerbasednavierstokesolver_amd import ops
    from oracle import transolver_oracle as orc
    B, N, heads, D, M = 2, 700, 8, 32, 64
    C = heads * D
    rng = np.random.default_rng(4242)
    g = lambda t: t.to(dev).contiguous()
    ws, bs = _r(rng, M, D, scale=D ** -0.5), 0.3 * _r(rng, M)
    wq, wk, wv = (_r(rng, D, D, scale=1.5 * D ** -0.5) for _ in range(3))
    temp = torch.tensor([0.1, 5.0, 0.01, 50.0, 0.5, 0.1000001, 4.9999995, 1.0])      # both clamp ends, inside and outside
    base = _r(rng, B, N, 2 * C)

    def run(xf, engine):
        spart, npart = ops.slice_scatter(g(xf), 2 * C, 0, g(xf), 2 * C, C, g(ws), g(bs), g(temp), B, N, heads, D, M, engine=engine)
        s, nrm = spart.sum(1).view(B, heads, M, D), npart.sum(1).view(B, heads, M)
        return s, nrm

    def run_y(xf, o, engine):
        return ops.deslice_fwd(g(xf), 2 * C, 0, g(o.float().reshape(B * heads, M, D).contiguous()), g(ws), g(bs), g(temp),
                               B, N, heads, D, M, engine=engine)

    def ref(xf):
        xd = xf.double()
        w, norm, s, tok = orc.slice_tokens(xd[..., :C], xd[..., C:], ws.double(), bs.double(), temp.double(), heads)
        o = orc.token_attention(tok, wq.double(), wk.double(), wv.double())
        return s, norm, o, orc.deslice(w, o)

    def err(a, r):
        a, r = a.double().cpu(), r.double().cpu()
        sc = 2.0 ** -round(float(torch.log2(r.abs().max().clamp_min(1e-300))))      # keep the norms inside fp64
        return float(((a - r) * sc).norm() / (r * sc).norm())

    scale = torch.from_numpy(np.exp2(rng.uniform(-40, 40, size=(B, N, 1))).astype(np.float32))
    spread = base.clone()
    spread[..., :C] *= 60.0                      # logits ~ 60 / 0.1: +-300 and more, far beyond exp's fp32 range
    zeros = base.clone()
    zeros[:, ::3] = 0.0                          # every third point: uniform weights, zero values
    zeros[..., 5] = 0.0
    for name, xf in (("range 2^+-40", base * scale), ("logit spread", spread), ("zeros", zeros)):
        s_r, n_r, o_r, y_r = ref(xf)
        e = {}
        for eng in ("f32", "split"):
            s, nrm = run(xf, eng)
            y = run_y(xf, o_r, eng)
            assert torch.isfinite(s).all() and torch.isfinite(nrm).all() and torch.isfinite(y).all(), (name, eng)
            e[eng] = (err(s, s_r), err(nrm, n_r), err(y, y_r))
        for q in range(3):
            assert e["split"][q] <= 2 * e["f32"][q] + 2e-7, (name, q, e)
            assert e["f32"][q] < 2e-5, (name, q, e)
    # backward kernel on the same wide-range operands (dS / dn / O from the oracle so that errors do not compound)
    xf = base * torch.from_numpy(np.exp2(rng.uniform(-12, 12, size=(B, N, 1))).astype(np.float32))
    dy = _r(rng, B, N, C)
    d = lambda t: t.double()
    rb = orc.slice_core_backward(d(xf[..., :C]), d(xf[..., C:]), d(dy), d(ws), d(bs), d(temp), d(wq), d(wk), d(wv), heads)
    o_ref = ref(xf)[2]
    f32 = lambda t, *shape: g(t.float().reshape(*shape))
    eb = {}
    for eng in ("f32", "split"):
        dxf, dws, dbs, dtemp = ops.slice_bwd_points(g(xf), g(dy), g(ws), g(bs), g(temp), f32(o_ref, B * heads, M, D),
                                                    f32(rb["ds"], B * heads, M, D), f32(rb["dn"], B * heads, M),
                                                    B, N, heads, D, M, engine=eng)
        eb[eng] = (err(dxf[..., :C], rb["dxm"]), err(dxf[..., C:], rb["dfm"]), err(dws, rb["dws"]), err(dbs, rb["dbs"]))
    for q in range(4):
        assert eb["split"][q] <= 2 * eb["f32"][q] + 1e-6, (q, eb)
    # +-inf / NaN: a poisoned x_mid row makes its point's weights NaN (all tokens of that (batch, head) and the point's own
    # output row), a poisoned fx_mid value its channel of that (batch, head)'s tokens: the same entries on both engines
    xp = base.clone()
    xp[0, 17, 3] = float("inf")                  # x_mid, head 0
    xp[1, 400, 2 * D + 1] = float("nan")         # x_mid, head 2
    xp[0, 300, C + 5 * D + 7] = -float("inf")    # fx_mid, head 5, channel 7
    o_fin = ref(base)[2]
    m = {}
    for eng in ("f32", "split"):
        s, nrm = run(xp, eng)
        y = run_y(xp, o_fin, eng)
        m[eng] = (torch.isfinite(s).cpu(), torch.isfinite(nrm).cpu(), torch.isfinite(y).cpu())
    for q in range(3):
        assert torch.equal(m["f32"][q], m["split"][q]), q
    bad_s, bad_y = ~m["split"][0], ~m["split"][2]
    assert bad_s[0, 0].all() and bad_s[1, 2].all() and bad_s[0, 5, :, 7].all()
    assert int(bad_s.sum()) == 2 * M * D + M                                # nothing else
    assert bad_y[0, 17, :D].all() and bad_y[1, 400, 2 * D:3 * D].all() and int(bad_y.sum()) == 2 * D


def _unplane(planes, rows, C, nt):
    """fp64 tensor [rows, C] held by a plane image [row][C/32][nt][32] bf16 (sum of the planes)"""
    p = planes.view(torch.bfloat16).view(rows, C // 32, nt, 32).double().sum(2)
    return p.reshape(rows, C)


@pytest.mark.parametrize("engine,nt", [("split", 3), ("bf16", 1)])
def test_operand_planes_interface(dev, engine, nt):
    """The producers of the conv operands emit the bf16 plane image directly (LayerNorm forward, slice backward + conv bias
    gradients) and the conv forward / backward consume it: same results as the fp32-tensor + pre-pass route."""
    from transformerbasednavierstokesolver_amd import ops
    B, H, W, heads, D, M = 2, 64, 64, 8, 32, 64
    C, N = heads * D, H * W
    assert ops.conv_planes_mask(B, H, W, C, engine) == 7 and ops.conv_planes_mask(B, H, W, 32, engine) != 7
    rng = np.random.default_rng(17)
    g = lambda t: t.to(dev).contiguous()
    tol = 1e-6 if nt == 3 else 4e-3
    # LayerNorm -> planes
    x, gam, bet = g(_r(rng, B * N, C) * 2 + 0.3), g(1 + 0.1 * _r(rng, C)), g(0.1 * _r(rng, C))
    y, mean, rstd = ops.layernorm_fwd(x, gam, bet)
    yp, mean2, rstd2 = ops.layernorm_fwd_planes(x, gam, bet, engine)
    assert torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert rel_l2(_unplane(yp, B * N, C, nt), y) < tol
    # conv forward from planes == conv forward from the fp32 tensor (bit-identical for the 3-plane split: same planes)
    wx, wf = g(_r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5)), g(_r(rng, C, C, 3, 3, scale=(9 * C) ** -0.5))
    bx, bf = g(0.1 * _r(rng, C)), g(0.1 * _r(rng, C))
    out_ref = ops.conv3x3x2_fwd(y.view(B, N, C), wx, bx, wf, bf, H, W, engine=engine)
    out_pl = ops.conv3x3x2_fwd_planes(yp, wx, bx, wf, bf, B, H, W, engine)
    assert torch.equal(out_pl, out_ref)
    # slice backward -> planes + conv bias gradients
    xf32, ws, bs, temp, wq, wk, wv, dy32 = _slice_inputs(B, N, heads, D, M, 5)
    xf, dy = g(xf32), g(dy32)
    spart, npart = ops.slice_scatter(xf, 2 * C, 0, xf, 2 * C, C, g(ws), g(bs), g(temp), B, N, heads, D, M)
    s_, nrm, o = ops.token_attn_fwd(spart, npart, g(wq), g(wk), g(wv))
    dopart, _ = ops.slice_scatter(xf, 2 * C, 0, dy, C, 0, g(ws), g(bs), g(temp), B, N, heads, D, M, want_norm=False)
    ds, dn, *_ = ops.token_attn_bwd(s_, nrm, g(wq), g(wk), g(wv), dopart)
    dxf, dws, dbs, dtemp = ops.slice_bwd_points(xf, dy, g(ws), g(bs), g(temp), o, ds, dn, B, N, heads, D, M)
    dxfp, dbx, dbf, dws2, dbs2, dtemp2 = ops.slice_bwd_points_planes(xf, dy, g(ws), g(bs), g(temp), o, ds, dn, nrm, B, N, heads,
                                                                     D, M, engine)
    assert rel_l2(_unplane(dxfp, B * N, 2 * C, nt), dxf.view(B * N, 2 * C)) < tol
    # the plain and the planes form are two instantiations of one kernel template: same sums, not the same instruction stream
    assert rel_l2(dws2, dws) < 1e-6 and rel_l2(dbs2, dbs) < 1e-6 and rel_l2(dtemp2, dtemp) < 1e-6
    colsum = dxf.view(B * N, 2 * C).double().sum(0)
    assert rel_l2(dbx, colsum[:C]) < 1e-5 and rel_l2(dbf, colsum[C:]) < 1e-5
    # conv backward from the two plane images == conv backward from fp32 tensors
    dxn_ref, dwx_ref, dbx_ref, dwf_ref, dbf_ref = ops.conv3x3x2_bwd(dxf, y.view(B, N, C), wx, wf, H, W, engine=engine)
    dxn, dwx, dwf = ops.conv3x3x2_bwd_planes(dxfp, yp, wx, wf, B, H, W, engine)
    assert rel_l2(dxn, dxn_ref) < 1e-6 and rel_l2(dwx, dwx_ref) < 1e-6 and rel_l2(dwf, dwf_ref) < 1e-6
    assert rel_l2(dbx, dbx_ref) < 1e-5 and rel_l2(dbf, dbf_ref) < 1e-5
    # accumulate flag
    into = tuple(torch.full_like(t, 0.25) for t in (dbx, dbf, dws, dbs, dtemp))
    ops.slice_bwd_points_planes(xf, dy, g(ws), g(bs), g(temp), o, ds, dn, nrm, B, N, heads, D, M, engine, into=into)
    for got, ref in zip(into, (dbx, dbf, dws, dbs, dtemp)):
        assert rel_l2(got - 0.25, ref) < 1e-5
