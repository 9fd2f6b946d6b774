"""End-to-end parity of the HIP path (Model / Attn / SOL / harness) against the golden vectors the
REFERENCE produced in the build container (tests/golden/G*.npz, made by oracle/make_golden.py).
Tolerances follow SURVEY.md §8c: fp32 forward rel-L2 <= 1e-5 per call, <= 2e-4 after 20 rollout
steps; gradients <= 1e-4 (<= 2e-3 for to_q/to_k)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _sample(t, n=257):
    f = torch.as_tensor(t).detach().double().cpu().numpy().ravel()
    stride = max(1, f.size // n)
    return f[::stride][:n]


def _grad_tol(key):
    return 2e-3 if ("to_q" in key or "to_k" in key) else 1e-4


def test_g1_tiny_model_forward_and_all_grads():
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss
    g = np.load(os.path.join(GOLDEN, "G1_tiny.npz"))
    sd = {k[3:]: g[k] for k in g.files if k.startswith("sd.")}
    m = harness.build_model(synth.TINY_CONFIG, sd, DEV)
    x, fx, y = (torch.from_numpy(g[k]).to(DEV) for k in ("x", "fx", "y"))
    pred = m(x, fx=fx)
    assert rel_l2(pred, g["pred.f64"]) < 1e-5
    loss = TestLoss(size_average=False)(pred.reshape(2, -1), y.reshape(2, -1))
    assert abs(loss.item() - float(g["loss.f64"])) < 1e-5 * abs(float(g["loss.f64"]))
    loss.backward()
    for k, p in m.named_parameters():
        if k == "placeholder":
            assert p.grad is None
            continue
        e = rel_l2(p.grad, g["grad.f64." + k])
        assert e < _grad_tol(k), (k, e)


def test_g1b_fx_none_and_time_input_branches():
    from transformerbasednavierstokesolver_amd import synth, harness
    g = np.load(os.path.join(GOLDEN, "G1b_tiny_branches.npz"))
    cfg = dict(synth.TINY_CONFIG, fun_dim=0, Time_Input=True, unified_pos=0)
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=12), DEV)
    x = torch.from_numpy(g["x"]).to(DEV)
    T = torch.from_numpy(g["T"]).reshape(-1, 1).to(DEV)
    with torch.no_grad():
        pred = m(x, None, T=T)
    assert rel_l2(pred, g["pred"]) < 2e-5


def test_g2_attention_module_ns_shape():
    from transformerbasednavierstokesolver_amd import synth
    from transformerbasednavierstokesolver_amd.model.Physics_Attention import Physics_Attention_Structured_Mesh_2D
    g = np.load(os.path.join(GOLDEN, "G2_attn_ns.npz"))
    cfg = synth.NS_CONFIG
    sd_all = synth.synth_state_dict(dict(cfg, n_layers=1), seed=21)
    pre = "blocks.0.Attn."
    C, h = cfg["n_hidden"], cfg["n_head"]
    a = Physics_Attention_Structured_Mesh_2D(C, heads=h, dim_head=C // h, slice_num=cfg["slice_num"], H=64, W=64)
    a.load_state_dict({k[len(pre):]: torch.from_numpy(v) for k, v in sd_all.items() if k.startswith(pre)}, strict=True)
    a = a.to(DEV)
    rng = np.random.default_rng(22)
    x = torch.from_numpy(rng.standard_normal((1, 4096, C)).astype(np.float32)).to(DEV).requires_grad_(True)
    gy = torch.from_numpy(rng.standard_normal((1, 4096, C)).astype(np.float32)).to(DEV)
    y = a(x)
    y.backward(gy)
    assert rel_l2(_sample(y), g["y.sample"]) < 1e-5
    assert abs(float(y.detach().double().norm()) - float(g["y.norm"])) < 1e-5 * float(g["y.norm"])
    assert rel_l2(_sample(x.grad), g["dx.sample"]) < 1e-4
    for k, p in a.named_parameters():
        assert rel_l2(_sample(p.grad, 129), g["grad.sample." + k]) < _grad_tol(k), k
        assert abs(float(p.grad.double().norm()) - float(g["grad.norm." + k])) < _grad_tol(k) * float(g["grad.norm." + k]), k


def test_g3_shipped_checkpoint_rollout_eager_and_graph():
    """The reference's own trained weights, 20-step prediction-feedback rollout."""
    from transformerbasednavierstokesolver_amd import synth, harness
    g = np.load(os.path.join(GOLDEN, "G3_shipped_rollout.npz"))
    ck = np.load(os.path.join(GOLDEN, "ckpt_ep400_sim100.npz"))
    m = harness.build_model(synth.NS_SMALL_CONFIG, {k: ck[k] for k in ck.files}, DEV).eval()
    pos, a, _ = synth.ns_batch(2, seed=31)
    x, fx = torch.from_numpy(pos).to(DEV), torch.from_numpy(a).to(DEV)
    fr = harness.rollout(m, x, fx, 20)
    for t, tol in ((0, 1e-5), (4, 1e-4), (9, 2e-4), (19, 2e-4)):
        # SURVEY §8c acceptance: stated tolerance OR "error vs the fp64 reference <= 4x the fp32
        # reference's own error" — on these out-of-distribution synthetic fields the reference's own
        # fp32 rollout is 4.0e-4 away from its fp64 rollout at frame 20 (both stored in the fixture).
        ref32 = rel_l2(g[f"frame{t + 1}.f32.sample"], g[f"frame{t + 1}.f64.sample"])
        tol = max(tol, 4 * ref32)
        assert rel_l2(_sample(fr[..., t]), g[f"frame{t + 1}.f64.sample"]) < tol, t
        assert abs(float(fr[..., t].double().norm()) - float(g[f"frame{t + 1}.f64.norm"])) < tol * float(g[f"frame{t + 1}.f64.norm"])
    gr = harness.GraphedRollout(m, x, fx)
    fr2 = gr.run(fx, 20)
    assert torch.equal(fr, fr2), "hipGraph replay must be bit-identical to the eager loop"


def test_g4_training_iteration_and_optimizer_step():
    from transformerbasednavierstokesolver_amd import synth, harness
    g = np.load(os.path.join(GOLDEN, "G4_train_iteration.npz"))
    cfg = synth.NS_SMALL_CONFIG
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=41), DEV).train()
    pos, a, u = synth.ns_batch(2, seed=42)
    x, fx, yy = (torch.from_numpy(t).to(DEV) for t in (pos, a, u))
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, epochs=5, steps_per_epoch=7)
    loss, full, pred = harness.train_iteration(m, x, fx, yy)
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    assert abs(full.item() - float(g["full"])) < 1e-5 * float(g["full"])
    assert rel_l2(_sample(pred), g["pred.sample"]) < 1e-5
    opt.zero_grad()
    loss.backward()
    for k, p in m.named_parameters():
        if k == "placeholder":
            assert p.grad is None
            continue
        assert rel_l2(_sample(p.grad, 65), g["grad.sample." + k]) < _grad_tol(k), k
        assert abs(float(p.grad.double().norm()) - float(g["grad.norm." + k])) < _grad_tol(k) * float(g["grad.norm." + k]), k
    opt.step()
    sched.step()
    assert abs(sched.get_last_lr()[0] - float(g["lr1"])) < 1e-12
    for k, p in m.named_parameters():
        assert abs(float(p.detach().double().norm()) - float(g["param1.norm." + k])) < 1e-5 * float(g["param1.norm." + k]) + 1e-12, k


@pytest.mark.parametrize("engine", ["f32", "split"])
def test_g5_full_ns_config_forward_backward(engine):
    """BASELINE configs[1] architecture: 8 layers, C=256, 8 heads, M=64, 64x64 — on BOTH fp32-accurate engines (exact
    fp32 MFMA, and the 6-term bf16 split that is the default) at the SAME fp32 tolerances."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss
    g = np.load(os.path.join(GOLDEN, "G5_full_ns.npz"))
    cfg = synth.NS_CONFIG
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=51), DEV, engine=engine)
    pos, a, u = synth.ns_batch(1, seed=52)
    x, fx, y = (torch.from_numpy(t).to(DEV) for t in (pos, a, u[..., :1]))
    pred = m(x, fx=fx)
    assert rel_l2(pred.reshape(-1), g["pred"]) < 1e-5
    loss = TestLoss(size_average=False)(pred.reshape(1, -1), y.reshape(1, -1))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    loss.backward()
    for k, p in m.named_parameters():
        if k == "placeholder":
            continue
        assert rel_l2(_sample(p.grad, 65), g["grad.sample." + k]) < _grad_tol(k), k
        assert abs(float(p.grad.double().norm()) - float(g["grad.norm." + k])) < _grad_tol(k) * float(g["grad.norm." + k]), k


def test_sol_wrapper_bptt_matches_oracle():
    """SOL_Transolver (n chained calls, predictions fed back, BPTT through all of them) vs the oracle."""
    from transformerbasednavierstokesolver_amd import synth
    from transformerbasednavierstokesolver_amd.model.SOL_Transolver_Structured_Mesh_2D import SOL_Transolver_Structured_Mesh_2D
    from oracle import transolver_oracle as orc
    cfg = dict(synth.TINY_CONFIG, out_dim=1, fun_dim=4)
    sd = synth.synth_state_dict(cfg, seed=61)
    kw = {k: cfg[k] for k in ("space_dim", "n_layers", "n_hidden", "dropout", "n_head", "Time_Input", "act", "mlp_ratio",
                              "fun_dim", "out_dim", "slice_num", "ref", "unified_pos", "H", "W")}
    sol = SOL_Transolver_Structured_Mesh_2D(**kw, step=1, look_ahead=3)
    sol.transolver_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sol = sol.to(DEV)
    rng = np.random.default_rng(62)
    N = cfg["H"] * cfg["W"]
    x = torch.from_numpy(rng.standard_normal((2, N, 2)).astype(np.float32))
    fx = torch.from_numpy(rng.standard_normal((2, N, 4)).astype(np.float32))
    u = sol(x.to(DEV), fx.to(DEV))
    u.square().sum().backward()
    sdo = orc.to_torch(sd, torch.float64, requires_grad=True)
    uo = orc.sol_forward(sdo, x.double(), fx.double(), cfg, 3)
    uo.square().sum().backward()
    assert rel_l2(u, uo) < 1e-5
    for k, p in sol.transolver_model.named_parameters():
        if k == "placeholder":
            continue
        assert rel_l2(p.grad, sdo[k].grad) < _grad_tol(k), k


@pytest.mark.parametrize("tag,c", [("s1n3", dict(fun_dim=4, out_dim=1, step=1, n=3, seed=71)),
                                   ("s2n2", dict(fun_dim=6, out_dim=2, step=2, n=2, seed=73))])
def test_g7_sol_wrapper_against_the_reference_made_fixture(tag, c):
    """HIP SOL wrapper vs G7 — output, loss and every gradient of the REFERENCE's SOL class (BPTT through n chained
    calls, model/SOL_Transolver_Structured_Mesh_2D.py:47-52)."""
    from transformerbasednavierstokesolver_amd import synth
    from transformerbasednavierstokesolver_amd.model.SOL_Transolver_Structured_Mesh_2D import SOL_Transolver_Structured_Mesh_2D
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss
    g = np.load(os.path.join(GOLDEN, "G7_sol_wrapper.npz"))
    cfg = dict(synth.TINY_CONFIG, fun_dim=c["fun_dim"], out_dim=c["out_dim"])
    kw = {k: cfg[k] for k in ("space_dim", "n_layers", "n_hidden", "dropout", "n_head", "Time_Input", "act", "mlp_ratio",
                              "fun_dim", "out_dim", "slice_num", "ref", "unified_pos", "H", "W")}
    sol = SOL_Transolver_Structured_Mesh_2D(**kw, step=c["step"], look_ahead=c["n"])
    sol.transolver_model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(cfg, seed=c["seed"]).items()},
                                         strict=True)
    sol = sol.to(DEV)
    x, fx, y = (torch.from_numpy(g[f"{tag}.{k}"]).to(DEV) for k in ("x", "fx", "y"))
    pred = sol(x, fx)
    loss = TestLoss(size_average=False)(pred.reshape(2, -1), y.reshape(2, -1))
    loss.backward()
    assert rel_l2(pred, g[f"{tag}.pred.f64"]) < 1e-5
    assert abs(loss.item() - float(g[f"{tag}.loss.f64"])) < 1e-5 * float(g[f"{tag}.loss.f64"])
    for k, p in sol.transolver_model.named_parameters():
        if k == "placeholder":
            assert p.grad is None
            continue
        assert rel_l2(p.grad, g[f"{tag}.grad.f64.{k}"]) < _grad_tol(k), k


def test_unrolled_lookahead_training_iteration_matches_oracle():
    """ns_vorticity_unrolling.py:225-244 through the SOL wrapper (look_ahead=2, BPTT through 2 calls)."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.model.SOL_Transolver_Structured_Mesh_2D import SOL_Transolver_Structured_Mesh_2D
    from oracle import transolver_oracle as orc
    cfg = dict(synth.TINY_CONFIG, out_dim=1, fun_dim=4)
    sd = synth.synth_state_dict(cfg, seed=81)
    kw = {k: cfg[k] for k in ("space_dim", "n_layers", "n_hidden", "dropout", "n_head", "Time_Input", "act", "mlp_ratio",
                              "fun_dim", "out_dim", "slice_num", "ref", "unified_pos", "H", "W")}
    sol = SOL_Transolver_Structured_Mesh_2D(**kw, step=1, look_ahead=1)
    sol.transolver_model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    sol = sol.to(DEV)
    rng = np.random.default_rng(82)
    N = cfg["H"] * cfg["W"]
    x = torch.from_numpy(rng.standard_normal((2, N, 2)).astype(np.float32))
    fx = torch.from_numpy(rng.standard_normal((2, N, 4)).astype(np.float32))
    yy = torch.from_numpy(rng.standard_normal((2, N, 4)).astype(np.float32))
    loss = harness.unrolled_train_iteration(sol, x.to(DEV), fx.to(DEV), yy.to(DEV), look_ahead=2)
    loss.backward()
    sdo = orc.to_torch(sd, torch.float64, requires_grad=True)
    lo = orc.unrolled_iteration_loss(sdo, x.double(), fx.double(), yy.double(), cfg, 2)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 1e-5 * abs(lo.item())
    for k, p in sol.transolver_model.named_parameters():
        if k != "placeholder":
            assert rel_l2(p.grad, sdo[k].grad) < _grad_tol(k), k
    cur = harness.LookAheadCurriculum(epochs=8, look_ahead=1, max_look_ahead=10)
    assert [cur.update(ep) for ep in range(8)] == [1, 1, 1, 1, 2, 2, 4, 8]


def test_darcy_iteration_matches_oracle():
    """exp_darcy.py:209-234 on a small grid: decode, rel-L2 + 0.1 x derivative loss, backward."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.utils.normalizer import UnitTransformer
    from oracle import transolver_oracle as orc
    s = 9
    cfg = synth.make_config(n_layers=2, n_hidden=32, n_head=4, slice_num=16, fun_dim=1, out_dim=1, ref=3, H=s, W=s)
    sd = synth.synth_state_dict(cfg, seed=91)
    m = harness.build_model(cfg, sd, DEV)
    rng = np.random.default_rng(92)
    x = torch.zeros(3, s * s, 2)
    fx = torch.from_numpy(rng.standard_normal((3, s * s)).astype(np.float32))
    y_raw = torch.from_numpy((2.0 + 0.5 * rng.standard_normal((3, s * s))).astype(np.float32))
    norm = UnitTransformer(y_raw)
    y = norm.encode(y_raw)
    dx = 1.0 / s
    out = m(x.to(DEV), fx=fx.to(DEV).unsqueeze(-1)).squeeze(-1)
    loss, l2, deriv = harness.darcy_loss(out, y.to(DEV), norm.to(DEV), dx, s)
    loss.backward()
    sdo = orc.to_torch(sd, torch.float64, requires_grad=True)
    oo = orc.model_forward(sdo, x.double(), fx.double().unsqueeze(-1), cfg).squeeze(-1)
    lo, l2o, dvo = orc.darcy_loss(oo, y.double(), norm.mean.cpu().double(), norm.std.cpu().double(), dx, s)
    lo.backward()
    assert abs(loss.item() - lo.item()) < 2e-5 * abs(lo.item())
    assert abs(deriv.item() - dvo.item()) < 2e-5 * abs(dvo.item())
    for k, p in m.named_parameters():
        if k != "placeholder":
            assert rel_l2(p.grad, sdo[k].grad) < _grad_tol(k), k


def test_bf16_compute_mode_full_ns_model():
    """BASELINE configs[2] numerics: the full NS model (8 layers, C=256) with every GEMM in bf16-compute
    mode against the fp64 reference output; SURVEY 8c tolerance for bf16 forward: rel-L2 <= 3e-2."""
    from transformerbasednavierstokesolver_amd import synth, harness, _lib
    g = np.load(os.path.join(GOLDEN, "G5_full_ns.npz"))
    cfg = synth.NS_CONFIG
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=51), DEV)
    pos, a, u = synth.ns_batch(1, seed=52)
    x, fx = torch.from_numpy(pos).to(DEV), torch.from_numpy(a).to(DEV)
    m.set_engine("bf16")
    pred = m(x, fx=fx)
    pred.square().sum().backward()
    e = rel_l2(pred.reshape(-1), g["pred"])
    assert 1e-5 < e < 3e-2, e          # really ran in reduced precision, and within the bf16 tolerance
    assert all(torch.isfinite(p.grad).all() for k, p in m.named_parameters() if p.grad is not None)


@pytest.mark.parametrize("tag", ["elas", "tiny"])
def test_g6_irregular_mesh_model(tag):
    """Transolver_Irregular_Mesh.Model (exp_elas.py geometry and a tiny unified_pos case)
    on the HIP kernels vs the reference-made fixture: Linear projections, unclamped temperature (values
    outside [0.1, 5] included), placeholder always added (and receiving a gradient)."""
    from test_oracle_golden import g6_inputs
    from transformerbasednavierstokesolver_amd.model_dict import get_model
    import types
    g = np.load(os.path.join(GOLDEN, "G6_irregular.npz"))
    cfg, sd, x, fx, gy = g6_inputs(tag)
    Model = get_model(types.SimpleNamespace(model="Transolver_Irregular_Mesh")).Model
    m = Model(space_dim=2, n_layers=cfg["n_layers"], n_hidden=cfg["n_hidden"], dropout=0.0, n_head=cfg["n_head"],
              Time_Input=False, mlp_ratio=cfg["mlp_ratio"], fun_dim=cfg["fun_dim"], out_dim=cfg["out_dim"],
              slice_num=cfg["slice_num"], ref=cfg["ref"], unified_pos=cfg["unified_pos"])
    res = m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    m = m.to(DEV)
    pred = m(torch.from_numpy(x).to(DEV), None if fx is None else torch.from_numpy(fx).to(DEV))
    pred.backward(torch.from_numpy(gy).to(DEV))
    assert rel_l2(_sample(pred), g[f"{tag}.pred.sample"]) < 1e-5
    for k, p in m.named_parameters():
        tol = _grad_tol(k)
        assert rel_l2(_sample(p.grad, 65), g[f"{tag}.grad.sample.{k}"]) < tol, k
        ref = float(g[f"{tag}.grad.norm.{k}"])
        assert abs(float(p.grad.double().norm()) - ref) < tol * ref, k


def test_full_bench_batch_consistency_and_determinism():
    """Size-independent properties at the BASELINE configs[1] batch (B=32, 64x64, C=256, M=64; 2 layers to
    keep it short): (1) the batch is a set of independent trajectories — outputs of the B=32 call equal the
    per-sample calls and the batch-summed-loss gradient equals the sum of per-sample gradients (different
    tile shapes are used for B=1, so equality is to fp32 rounding, not bitwise); (2) no float atomics
    anywhere: two identical runs give bitwise identical outputs and gradients."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss
    cfg = dict(synth.NS_CONFIG, n_layers=2)
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=131), DEV)
    pos, a, u = synth.ns_batch(32, seed=132)
    x, fx, y = (torch.from_numpy(t).to(DEV) for t in (pos, a, u[..., :1]))
    loss_fn = TestLoss(size_average=False)

    def run(xs, fs, ys):
        m.zero_grad(set_to_none=True)
        pred = m(xs, fx=fs)
        loss_fn(pred.reshape(xs.shape[0], -1), ys.reshape(xs.shape[0], -1)).backward()
        return pred.detach().clone(), {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    p1, g1 = run(x, fx, y)
    p2, g2 = run(x, fx, y)
    assert torch.equal(p1, p2) and all(torch.equal(g1[k], g2[k]) for k in g1), "run-to-run results must be bitwise identical"
    acc = None
    for i in (0, 13, 31):
        pi, gi = run(x[i:i + 1], fx[i:i + 1], y[i:i + 1])
        assert rel_l2(pi, p1[i:i + 1]) < 1e-5, i      # fp32 forward tolerance
    # gradient additivity over a split of the batch into 4 shards of 8 (what DDP relies on)
    for s in range(4):
        _, gs = run(x[8 * s:8 * s + 8], fx[8 * s:8 * s + 8], y[8 * s:8 * s + 8])
        acc = gs if acc is None else {k: acc[k] + gs[k] for k in acc}
    for k in g1:
        assert rel_l2(acc[k], g1[k]) < (2e-3 if ("to_q" in k or "to_k" in k) else 1e-4), k


def test_frozen_weight_packs_follow_weight_updates():
    """ops.weights_frozen(): (a) a scoped rollout (conv weights packed once per layer) is bit-identical to
    unscoped model calls (packed per call); (b) a GraphedRollout whose captured step only references the packs
    still sees parameters changed after the capture, even through `.data` (which bumps no version counter):
    run() refreshes the packs in place."""
    from transformerbasednavierstokesolver_amd import synth, harness, ops
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=61), DEV).eval()
    pos, a, _ = synth.ns_batch(2, seed=62)
    x, fx = torch.from_numpy(pos).to(DEV), torch.from_numpy(a).to(DEV)

    def unscoped(n):
        frames, w = [], fx
        with torch.no_grad():
            for _ in range(n):
                assert not ops._frozen
                im = m(x, fx=w)
                frames.append(im)
                w = torch.cat((w[..., 1:], im), dim=-1)
        return torch.cat(frames, -1)

    assert torch.equal(harness.rollout(m, x, fx, 4), unscoped(4))
    gr = harness.GraphedRollout(m, x, fx)
    assert len(gr.packs.packs) == cfg["n_layers"]
    assert torch.equal(gr.run(fx, 4), unscoped(4))
    with torch.no_grad():
        for blk in m.blocks:
            blk.Attn.in_project_x.weight.data.mul_(1.25)
            blk.Attn.in_project_fx.weight.data.add_(0.01)
    after = unscoped(4)
    assert not torch.equal(after[..., 0], gr.im[..., 0])
    assert torch.equal(gr.run(fx, 4), after)


def test_empty_batch_forward_and_backward():
    """B = 0 (the reference's PyTorch ops accept it): the output is [0,N,out_dim], backward runs and every
    parameter on the path gets an exactly-zero gradient — reductions over an empty set are zero-filled by
    libpa2d, maps are no-ops, no kernel is launched with an empty grid."""
    from transformerbasednavierstokesolver_amd import synth, harness
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=71), DEV).train()
    pos, a, _ = synth.ns_batch(1, seed=72)
    x, fx = torch.from_numpy(pos).to(DEV), torch.from_numpy(a).to(DEV)
    out = m(x[:0], fx=fx[:0])
    assert out.shape == (0, cfg["H"] * cfg["W"], cfg["out_dim"])
    out.sum().backward()
    seen = 0
    for k, p in m.named_parameters():
        if k == "placeholder":
            continue
        assert p.grad is not None and p.grad.shape == p.shape and not p.grad.any(), k
        seen += 1
    assert seen == len(synth.state_dict_spec(cfg)) - 1
    with torch.no_grad():
        assert m(x[:0], fx=fx[:0]).shape[0] == 0


def test_c_abi_error_codes_on_device():
    """Error behaviour of the boundary: contract violations come back as PA2D_ERR_* codes (and RuntimeError in the
    ctypes layer) BEFORE anything is launched."""
    from transformerbasednavierstokesolver_amd import _lib, ops
    L = _lib.load()
    t = torch.zeros(64 * 64, device=DEV)
    ptr = t.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    # K not a multiple of 4 -> ARG
    assert L.pa2d_gemm_bias_act_fwd(ptr, 6, ptr, 6, 0, 0, 8, ptr, 8, 0, 8, 0, 0, 0, 16, 8, 6, 0, 0, st) == 1001
    # unknown engine id -> ARG
    assert L.pa2d_gemm_bias_act_fwd(ptr, 8, ptr, 8, 0, 0, 8, ptr, 8, 0, 8, 0, 0, 0, 16, 8, 8, 0, 7, st) == 1001
    # LayerNorm width not a multiple of 4 -> UNSUPPORTED
    assert L.pa2d_layernorm_fwd(ptr, ptr, ptr, ptr, ptr, ptr, 4, 30, 1e-5, st) == 1002
    # more slices than the token kernel holds -> UNSUPPORTED
    assert L.pa2d_token_attn_fwd(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, 1, 1, 129, 8, st) == 1002
    # head dim outside {8,16,32,64} -> ARG/UNSUPPORTED
    for engine in (0, 1):
        assert L.pa2d_slice_scatter(ptr, 12, ptr, 12, ptr, ptr, ptr, ptr, ptr, 1, 16, 1, 12, 8, 1, engine, st, 0, 0) in (1001, 1002)
    # unknown slice engine id -> ARG
    assert L.pa2d_slice_scatter(ptr, 16, ptr, 16, ptr, ptr, ptr, ptr, ptr, 1, 16, 1, 16, 8, 1, 5, st, 0, 0) == 1001
    assert L.pa2d_deslice_fwd(ptr, 16, ptr, ptr, ptr, ptr, ptr, 16, 1, 16, 1, 16, 8, 1, -1, st, 0, 0) == 1001
    # workspace too small -> WORKSPACE
    assert L.pa2d_gemm_bwd_weight(ptr, 8, ptr, 8, ptr, ptr, ptr, 16, 64, 8, 8, 0, 0, st) == 1003
    for engine in (0, 1, 2):
        assert L.pa2d_conv3x3x2_fwd(ptr, ptr, ptr, ptr, ptr, ptr, 0, ptr, 16, 1, 4, 4, 16, engine, st, 0, 0) == 1003
    # a problem whose operand would exceed the 4 GiB buffer-descriptor extent -> UNSUPPORTED (nothing launched)
    assert L.pa2d_gemm_bias_act_fwd(ptr, 512, ptr, 512, 0, 0, 512, ptr, 512, 0, 512, 0, 0, 0, 2200000, 512, 512, 0, 0, st) == 1002
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="PA2D_ERR"):
        _lib.check(1002, "probe")
    with pytest.raises(TypeError):
        ops.linear_fwd(t.view(64, 64).double(), t.view(64, 64))
    with pytest.raises(ValueError):
        ops.linear_fwd(t.view(64, 64).t(), t.view(64, 64))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.linear_fwd(torch.zeros(4, 4), torch.zeros(4, 4))


def test_two_models_on_different_engines_coexist():
    """The engine is a per-call argument (no process state): a model on the exact engine and one on the bf16-compute
    engine interleave in one process — inside ONE weights_frozen scope, so their weight packs (different layouts)
    must not cross — and each reproduces its own solo result bit for bit."""
    from transformerbasednavierstokesolver_amd import synth, harness, ops
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=81)
    pos, a, _ = synth.ns_batch(1, seed=82)
    x, fx = torch.from_numpy(pos).to(DEV), torch.from_numpy(a).to(DEV)
    models = {e: harness.build_model(cfg, sd, DEV, engine=e).eval() for e in ("f32", "split", "bf16")}
    with torch.no_grad():
        solo = {e: m(x, fx=fx) for e, m in models.items()}
        with ops.weights_frozen():
            mixed = {}
            for _ in range(2):                        # second round reuses the packs made in the first
                for e, m in models.items():
                    mixed[e] = m(x, fx=fx)
    for e in models:
        assert torch.equal(mixed[e], solo[e]), e
    assert rel_l2(solo["split"], solo["f32"]) < 1e-5 and 1e-5 < rel_l2(solo["bf16"], solo["f32"]) < 3e-2
    assert harness.model_engine(models["bf16"]) == ops.ENGINE_BF16
    # a graphed rollout keeps ITS model's engine whatever other models run in between
    gr = harness.GraphedRollout(models["split"], x, fx)
    ref = harness.rollout(models["split"], x, fx, 2)
    with torch.no_grad():
        models["bf16"](x, fx=fx)
    assert torch.equal(gr.run(fx, 2), ref)
