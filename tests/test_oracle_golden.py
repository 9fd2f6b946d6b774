"""CPU: the oracle (oracle/transolver_oracle.py) against the golden vectors the REFERENCE produced
(tests/golden/*.npz, written by oracle/make_golden.py in the build container).  This is what pins
the oracle; the GPU tests then pin the HIP path against the same vectors and against the oracle."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2
from oracle import transolver_oracle as orc
from transformerbasednavierstokesolver_amd import synth


def _sample(t, n=257):
    f = torch.as_tensor(t).detach().double().numpy().ravel()
    stride = max(1, f.size // n)
    return f[::stride][:n]


@pytest.mark.parametrize("tag,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-5)])
def test_g1_tiny_forward_loss_and_all_grads(tag, dtype, tol):
    g = np.load(os.path.join(GOLDEN, "G1_tiny.npz"))
    cfg = synth.TINY_CONFIG
    sd = orc.to_torch({k[3:]: g[k] for k in g.files if k.startswith("sd.")}, dtype, requires_grad=True)
    x, fx, y = (torch.from_numpy(g[k]).to(dtype) for k in ("x", "fx", "y"))
    pred = orc.model_forward(sd, x, fx, cfg)
    assert rel_l2(pred, g[f"pred.{tag}"]) < tol
    loss = orc.rel_l2(pred.reshape(2, -1), y.reshape(2, -1))
    assert abs(loss.item() - float(g[f"loss.{tag}"])) < 50 * tol * abs(float(g[f"loss.{tag}"]))
    loss.backward()
    assert sd["placeholder"].grad is None            # unused on the fx-given path (…_2D.py:205-210)
    for k in sd:
        if k != "placeholder":
            assert rel_l2(sd[k].grad, g[f"grad.{tag}.{k}"]) < 50 * tol, k


def test_g1_clamp_mask_is_pinned():
    """temperatures outside [0.1, 5] (Physics_Attention.py:99) get exactly zero gradient."""
    g = np.load(os.path.join(GOLDEN, "G1_tiny.npz"))
    t = g["sd.blocks.0.Attn.temperature"].reshape(-1)
    gr = g["grad.f64.blocks.0.Attn.temperature"].reshape(-1)
    outside = (t < 0.1) | (t > 5.0)
    assert outside.any() and (~outside).any()
    assert np.all(gr[outside] == 0) and np.all(gr[~outside] != 0)


def test_g1b_fx_none_and_time_branches():
    g = np.load(os.path.join(GOLDEN, "G1b_tiny_branches.npz"))
    cfg = dict(synth.TINY_CONFIG, fun_dim=0, Time_Input=True, unified_pos=0)
    sd = orc.to_torch(synth.synth_state_dict(cfg, seed=12))
    pred = orc.model_forward(sd, torch.from_numpy(g["x"]), None, cfg, T=torch.from_numpy(g["T"]).reshape(-1, 1))
    assert rel_l2(pred, g["pred"]) < 2e-5


def test_g2_attention_module_ns_shape_forward():
    g = np.load(os.path.join(GOLDEN, "G2_attn_ns.npz"))
    cfg = synth.NS_CONFIG
    sd = orc.to_torch({k: v for k, v in synth.synth_state_dict(dict(cfg, n_layers=1), seed=21).items()
                       if k.startswith("blocks.0.Attn.")}, torch.float64)
    rng = np.random.default_rng(22)
    x = torch.from_numpy(rng.standard_normal((1, 4096, cfg["n_hidden"])).astype(np.float32)).double()
    y = orc.physics_attention(x, sd, "blocks.0.Attn.", 64, 64, cfg["n_head"])
    assert rel_l2(_sample(y), g["y.sample"]) < 1e-12
    assert abs(float(y.norm()) - float(g["y.norm"])) < 1e-10 * float(g["y.norm"])


def test_g3_shipped_checkpoint_rollout_first_frames():
    """The reference's own trained weights (checkpoints/ep400_sim100.pt as npz data)."""
    g = np.load(os.path.join(GOLDEN, "G3_shipped_rollout.npz"))
    ck = np.load(os.path.join(GOLDEN, "ckpt_ep400_sim100.npz"))
    sd = orc.to_torch({k: ck[k] for k in ck.files}, torch.float64)
    assert len(sd) == 169 and sum(v.numel() for v in sd.values()) == 714753
    pos, a, _ = synth.ns_batch(2, seed=31)
    fr = orc.rollout(sd, torch.from_numpy(pos).double(), torch.from_numpy(a).double(), synth.NS_SMALL_CONFIG, 5)
    for t in (0, 4):
        assert rel_l2(_sample(fr[..., t]), g[f"frame{t + 1}.f64.sample"]) < 1e-11


def test_g5_full_ns_forward():
    g = np.load(os.path.join(GOLDEN, "G5_full_ns.npz"))
    cfg = synth.NS_CONFIG
    sd = orc.to_torch(synth.synth_state_dict(cfg, seed=51), torch.float64)
    pos, a, _ = synth.ns_batch(1, seed=52)
    with torch.no_grad():
        pred = orc.model_forward(sd, torch.from_numpy(pos).double(), torch.from_numpy(a).double(), cfg)
    assert rel_l2(pred.reshape(-1), g["pred"]) < 1e-12


def test_hand_derived_backward_matches_autograd():
    """SURVEY Appendix A.2 formulas (used to test the HIP backward stage by stage) vs autograd."""
    B, N, h, D, M = 2, 37, 4, 8, 12
    C = h * D
    rng = np.random.default_rng(3)
    r = lambda *s: torch.from_numpy(rng.standard_normal(s)).double()
    xm, fm, dy = r(B, N, C), r(B, N, C), r(B, N, C)
    ws, bs = r(M, D) * 0.4, r(M) * 0.3
    temp = torch.tensor([0.03, 0.5, 7.0, 1.5], dtype=torch.float64)
    wq, wk, wv = r(D, D) * 0.5, r(D, D) * 0.5, r(D, D) * 0.5
    leaves = [t.requires_grad_(True) for t in (xm, fm, ws, bs, temp, wq, wk, wv)]
    w, norm, s, tok = orc.slice_tokens(xm, fm, ws, bs, temp, h)
    y = orc.deslice(w, orc.token_attention(tok, wq, wk, wv))
    grads = torch.autograd.grad(y, leaves, dy)
    ref = orc.slice_core_backward(xm.detach(), fm.detach(), dy, ws.detach(), bs.detach(), temp.detach(),
                                  wq.detach(), wk.detach(), wv.detach(), h)
    for name, gr in zip(("dxm", "dfm", "dws", "dbs", "dtemperature", "dwq", "dwk", "dwv"), grads):
        assert rel_l2(ref[name].reshape(gr.shape), gr) < 1e-12, name


G6_CASES = {"elas": (dict(n_layers=3, n_hidden=128, n_head=8, slice_num=64, fun_dim=0, out_dim=1, unified_pos=0), 2, 972, 111),
            "tiny": (dict(n_layers=2, n_hidden=32, n_head=4, slice_num=12, fun_dim=3, out_dim=2, unified_pos=1, ref=3,
                          mlp_ratio=2), 2, 45, 113)}


def g6_inputs(tag):
    kw, B, N, seed = G6_CASES[tag]
    cfg = synth.make_config(**kw)
    sd = synth.synth_irregular_state_dict(cfg, seed=seed)
    rng = np.random.default_rng(seed + 1)
    x = rng.uniform(0, 1, (B, N, 2)).astype(np.float32)
    fx = rng.standard_normal((B, N, cfg["fun_dim"])).astype(np.float32) if cfg["fun_dim"] else None
    gy = rng.standard_normal((B, N, cfg["out_dim"])).astype(np.float32)
    return cfg, sd, x, fx, gy


@pytest.mark.parametrize("tag", ["elas", "tiny"])
def test_g6_irregular_mesh_family(tag):
    """oracle of model/Transolver_Irregular_Mesh.py vs the reference-made fixture."""
    g = np.load(os.path.join(GOLDEN, "G6_irregular.npz"))
    cfg, sd, x, fx, gy = g6_inputs(tag)
    sdo = orc.to_torch(sd, torch.float64, requires_grad=True)
    pred = orc.model_forward_irregular(sdo, torch.from_numpy(x).double(), None if fx is None else torch.from_numpy(fx).double(), cfg)
    pred.backward(torch.from_numpy(gy).double())
    assert rel_l2(_sample(pred), g[f"{tag}.pred.sample"]) < 1e-12
    for k in sd:
        assert abs(float(sdo[k].grad.norm()) - float(g[f"{tag}.grad.norm.{k}"])) < 1e-9 * float(g[f"{tag}.grad.norm.{k}"]) + 1e-300, k


G7_CASES = {"s1n3": dict(fun_dim=4, out_dim=1, step=1, n=3, seed=71), "s2n2": dict(fun_dim=6, out_dim=2, step=2, n=2, seed=73)}


@pytest.mark.parametrize("tag", list(G7_CASES))
@pytest.mark.parametrize("dt,dtype,tol", [("f64", torch.float64, 1e-12), ("f32", torch.float32, 2e-5)])
def test_g7_sol_wrapper_bptt(tag, dt, dtype, tol):
    """oracle.sol_forward vs the fixture the reference's SOL_Transolver_Structured_Mesh_2D produced
    (model/SOL_Transolver_Structured_Mesh_2D.py:47-52): n chained calls, loss on the last prediction, BPTT."""
    g = np.load(os.path.join(GOLDEN, "G7_sol_wrapper.npz"))
    c = G7_CASES[tag]
    cfg = dict(synth.TINY_CONFIG, fun_dim=c["fun_dim"], out_dim=c["out_dim"])
    sd = orc.to_torch(synth.synth_state_dict(cfg, seed=c["seed"]), dtype, requires_grad=True)
    x, fx, y = (torch.from_numpy(g[f"{tag}.{k}"]).to(dtype) for k in ("x", "fx", "y"))
    pred = orc.sol_forward(sd, x, fx, cfg, c["n"], step=c["step"])
    assert rel_l2(pred, g[f"{tag}.pred.{dt}"]) < tol
    loss = orc.rel_l2(pred.reshape(2, -1), y.reshape(2, -1))
    assert abs(loss.item() - float(g[f"{tag}.loss.{dt}"])) < 50 * tol * abs(float(g[f"{tag}.loss.{dt}"]))
    loss.backward()
    for k in sd:
        if k != "placeholder":
            assert rel_l2(sd[k].grad, g[f"{tag}.grad.{dt}.{k}"]) < 50 * tol, k
