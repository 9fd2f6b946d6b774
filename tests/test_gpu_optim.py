"""fused AdamW (+ OneCycleLR-driven lr/beta1, grad-norm clip) and fused rel-L2 loss
against torch.optim.AdamW / clip_grad_norm_ / the reference TestLoss formula on the GPU."""
import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _toy():
    torch.manual_seed(5)
    m = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.GELU(), torch.nn.Linear(53, 3)).to(DEV)
    m.unused = torch.nn.Parameter(torch.rand(7, device=DEV))        # never receives a gradient
    return m


@pytest.mark.parametrize("clip", [None, 0.05])
def test_fused_adamw_matches_torch_with_onecycle(clip):
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    a, b = _toy(), _toy()
    oa = torch.optim.AdamW(a.parameters(), lr=1e-3, weight_decay=1e-5)
    ob = FusedAdamW(b.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=clip)
    sa = torch.optim.lr_scheduler.OneCycleLR(oa, max_lr=1e-3, total_steps=12)
    sb = torch.optim.lr_scheduler.OneCycleLR(ob, max_lr=1e-3, total_steps=12)
    g = torch.Generator(device=DEV).manual_seed(1)
    for it in range(6):
        x = torch.randn(64, 37, device=DEV, generator=g)
        for m, o, s in ((a, oa, sa), (b, ob, sb)):
            o.zero_grad()
            (m(x).square().sum() * 3.0).backward()
            if o is oa and clip is not None:
                torch.nn.utils.clip_grad_norm_(a.parameters(), clip)
            o.step()
            s.step()
    for (ka, pa), (kb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert ka == kb
        assert rel_l2(pb, pa) < 2e-6, ka
    assert b.unused.grad is None and torch.equal(a.unused, b.unused)
    assert abs(sa.get_last_lr()[0] - sb.get_last_lr()[0]) < 1e-15
    assert oa.param_groups[0]["betas"][0] == ob.param_groups[0]["betas"][0]


@pytest.mark.parametrize("B,L", [(2, 60), (32, 4096), (5, 40960)])
def test_fused_rel_l2_loss_and_gradient(B, L):
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss, FusedTestLoss
    g = torch.Generator(device=DEV).manual_seed(B + L)
    pred = torch.randn(B, L, device=DEV, generator=g)
    y = torch.randn(B, L, device=DEV, generator=g)
    for avg in (False, True):
        p1 = pred.clone().requires_grad_(True)
        p2 = pred.double().clone().requires_grad_(True)
        l1 = FusedTestLoss(size_average=avg)(p1, y)
        l2 = TestLoss(size_average=avg)(p2, y.double())
        l1.backward()
        l2.backward()
        assert abs(l1.item() - l2.item()) < 2e-6 * abs(l2.item())
        assert rel_l2(p1.grad, p2.grad) < 5e-6


def test_training_step_with_fused_optimizer_and_loss_matches_reference_fixture():
    """G4 again (the reference's exp_ns iteration + AdamW/OneCycle step), now through FusedAdamW and
    FusedTestLoss: parameters after the step must match the reference-made fixture."""
    import os
    from conftest import GOLDEN
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    g = np.load(os.path.join(GOLDEN, "G4_train_iteration.npz"))
    cfg = synth.NS_SMALL_CONFIG
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=41), DEV).train()
    pos, a, u = synth.ns_batch(2, seed=42)
    x, fx, yy = (torch.from_numpy(t).to(DEV) for t in (pos, a, u))
    opt = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, epochs=5, steps_per_epoch=7)
    loss, full, _ = harness.train_iteration(m, x, fx, yy, loss_fn=FusedTestLoss(size_average=False))
    assert abs(loss.item() - float(g["loss"])) < 1e-5 * float(g["loss"])
    opt.zero_grad()
    loss.backward()
    opt.step()
    sched.step()
    for k, p in m.named_parameters():
        ref = float(g["param1.norm." + k])
        assert abs(float(p.detach().double().norm()) - ref) < 1e-5 * ref + 1e-12, k
        s = p.detach().double().cpu().numpy().ravel()
        stride = max(1, s.size // 65)
        assert rel_l2(s[::stride][:65], g["param1.sample." + k]) < 1e-5, k


def test_multi_step_training_trajectory_matches_oracle():
    """4 consecutive exp_ns-style iterations (3 teacher-forced calls each) on the HIP path with FusedAdamW +
    OneCycleLR + FusedTestLoss vs the fp64 oracle trained by torch.optim.AdamW on the same data: parameters
    and losses must track each other step after step (flat buffers, weight re-packs and moments are reused
    across steps, which single-step tests cannot see)."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    from oracle import transolver_oracle as orc
    cfg = dict(synth.TINY_CONFIG, out_dim=1, fun_dim=4)
    sd = synth.synth_state_dict(cfg, seed=121)
    m = harness.build_model(cfg, sd, DEV).train()
    N = cfg["H"] * cfg["W"]
    rng = np.random.default_rng(122)
    x = torch.from_numpy(rng.standard_normal((3, N, 2)).astype(np.float32))
    fx = torch.from_numpy(rng.standard_normal((3, N, 4)).astype(np.float32))
    yy = torch.from_numpy(rng.standard_normal((3, N, 3)).astype(np.float32))
    opt = FusedAdamW(m.parameters(), lr=2e-3, weight_decay=1e-5, max_grad_norm=0.5)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=2e-3, total_steps=10)
    sdo = orc.to_torch(sd, torch.float64, requires_grad=True)
    live = [k for k in sdo if k != "placeholder"]
    oo = torch.optim.AdamW([sdo[k] for k in live], lr=2e-3, weight_decay=1e-5)
    so = torch.optim.lr_scheduler.OneCycleLR(oo, max_lr=2e-3, total_steps=10)
    for it in range(4):
        loss, _ = harness.train_step(m, opt, sched, x.to(DEV), fx.to(DEV), yy.to(DEV), grad_sync=opt.sync,
                                     loss_fn=FusedTestLoss(size_average=False))
        oo.zero_grad()
        lo, _, _, grads = orc.train_iteration(sdo, x.double(), fx.double(), yy.double(), cfg)
        for k in live:
            sdo[k].grad = grads[k]
        torch.nn.utils.clip_grad_norm_([sdo[k] for k in live], 0.5)
        oo.step()
        so.step()
        assert abs(loss.item() - lo.item()) < 2e-5 * abs(lo.item()), it
    for k, p in m.named_parameters():
        assert rel_l2(p, sdo[k]) < 2e-5, k


def test_graphed_training_step_is_bit_identical_to_eager():
    """harness.GraphedTrainStep (hipGraph of zero-grad + 10 forward calls + backward, eager fused optimizer)
    vs harness.train_step on two identically initialised models: losses and parameters bitwise equal over
    3 iterations with changing data (OneCycle lr/beta1 included)."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=141)
    models, opts, scheds = [], [], []
    for _ in range(2):
        m = harness.build_model(cfg, sd, DEV).train()
        o = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        models.append(m); opts.append(o)
        scheds.append(torch.optim.lr_scheduler.OneCycleLR(o, max_lr=1e-3, total_steps=8))
    batches = []
    for it in range(3):
        pos, a, u = synth.ns_batch(2, seed=150 + it)
        batches.append(tuple(torch.from_numpy(t).to(DEV) for t in (pos, a, u)))
    lf = FusedTestLoss(size_average=False)
    graphed = harness.GraphedTrainStep(models[1], opts[1], scheds[1], *batches[0], loss_fn=lf)
    for x, fx, yy in batches:
        le, _ = harness.train_step(models[0], opts[0], scheds[0], x, fx, yy, grad_sync=opts[0].sync, loss_fn=lf)
        lg, _ = graphed(x, fx, yy)
        assert torch.equal(le, lg)
    for (k, pe), (_, pg) in zip(models[0].named_parameters(), models[1].named_parameters()):
        assert torch.equal(pe, pg), k


def test_time_folded_iteration_matches_sequential_calls():
    """harness.train_iteration(fold_time=True) (ONE model call on the 10*B teacher-forced windows) vs the
    sequential loop of exp_ns.py:199-207 on the same model: predictions and per-call loss agree to fp32
    rounding (the per-sample arithmetic is identical; only the chunk partial sums of the slice scatter can
    regroup), gradients agree to the fp32 summation-order tolerance of the weight-gradient reductions, and
    one full training step leaves the same parameters."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=3)
    sd = synth.synth_state_dict(cfg, seed=171)
    pos, a, u = synth.ns_batch(3, seed=172)
    x, fx, yy = (torch.from_numpy(t).to(DEV) for t in (pos, a, u))
    lf = FusedTestLoss(size_average=False)
    grads, outs = [], []
    for fold in (False, True):
        m = harness.build_model(cfg, sd, DEV).train()
        loss, full, pred = harness.train_iteration(m, x, fx, yy, loss_fn=lf, fold_time=fold)
        loss.backward()
        grads.append({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None})
        outs.append((loss.detach(), full, pred.detach()))
    assert rel_l2(outs[1][2], outs[0][2]) < 2e-6
    assert abs(float(outs[1][0]) - float(outs[0][0])) <= 2e-6 * abs(float(outs[0][0]))
    assert abs(float(outs[1][1]) - float(outs[0][1])) <= 2e-6 * abs(float(outs[0][1]))
    assert grads[0].keys() == grads[1].keys()
    for k in grads[0]:
        assert rel_l2(grads[1][k], grads[0][k]) < 2e-5, k
    # a full optimizer step from both paths
    params = []
    for fold in (False, True):
        m = harness.build_model(cfg, sd, DEV).train()
        o = FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, max_grad_norm=1.0)
        harness.train_step(m, o, None, x, fx, yy, grad_sync=o.sync, loss_fn=lf, fold_time=fold)
        params.append(dict(m.named_parameters()))
    for k in params[0]:
        assert rel_l2(params[1][k], params[0][k]) < 2e-5, k


def test_time_folding_splits_into_groups_under_the_descriptor_limit(monkeypatch):
    """fold_time must never build a call whose widest activation exceeds the 4 GiB buffer-descriptor extent: with
    the limit lowered so that only 4 of the 10 calls fit, the iteration runs as 3 groups (4+4+2) and still matches
    the sequential loop."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=181)
    pos, a, u = synth.ns_batch(2, seed=182)
    x, fx, yy = (torch.from_numpy(t).to(DEV) for t in (pos, a, u))
    m = harness.build_model(cfg, sd, DEV, engine="f32").train()
    assert harness._fold_group(m, 2, 4096, 10) == 10
    assert harness._fold_group(harness.build_model(synth.NS_CONFIG, None, "cpu", engine="f32"), 64, 4096, 10) == 7   # B=64, C=256
    # 3-plane bf16 image of the 2C-wide gradient on the split engine: 6 B per element
    assert harness._fold_group(harness.build_model(synth.NS_CONFIG, None, "cpu", engine="split"), 64, 4096, 10) == 5
    monkeypatch.setattr(harness, "FOLD_MAX_BYTES", 4 * 2 * 64 * (4 * 2 * 4096) + 100)
    assert harness._fold_group(m, 2, 4096, 10) == 4
    calls = []
    orig = m.forward
    m.forward = lambda *a_, **k_: (calls.append(a_[0].shape[0]), orig(*a_, **k_))[1]
    lf = FusedTestLoss(size_average=False)
    loss_f, full_f, pred_f = harness.train_iteration(m, x, fx, yy, loss_fn=lf, fold_time=True)
    assert calls == [8, 8, 4]
    m.forward = orig
    loss_s, full_s, pred_s = harness.train_iteration(m, x, fx, yy, loss_fn=lf)
    assert rel_l2(pred_f.detach(), pred_s.detach()) < 2e-6
    assert abs(float(loss_f.detach()) - float(loss_s.detach())) < 2e-6 * abs(float(loss_s.detach()))


def test_fused_adamw_state_dict_round_trips_moments_and_step():
    """save -> load into a fresh optimizer -> step == uninterrupted run, and == torch.optim.AdamW (whose state_dict
    round-trips the moments and step count too)."""
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    g = torch.Generator(device=DEV).manual_seed(9)
    xs = [torch.randn(32, 37, device=DEV, generator=g) for _ in range(5)]

    def run(m, o, batches):
        for x in batches:
            o.zero_grad()
            m(x).square().sum().backward()
            o.step()

    a, b, c = _toy(), _toy(), _toy()
    oa = FusedAdamW(a.parameters(), lr=3e-3, weight_decay=1e-2)
    run(a, oa, xs)                                           # uninterrupted
    ob = FusedAdamW(b.parameters(), lr=3e-3, weight_decay=1e-2)
    run(b, ob, xs[:3])
    saved = ob.state_dict()
    assert int(saved["fused"]["steps"]) == 3 and float(saved["fused"]["exp_avg"].abs().sum()) > 0
    b2 = _toy()
    b2.load_state_dict(b.state_dict())
    ob2 = FusedAdamW(b2.parameters(), lr=3e-3, weight_decay=1e-2)
    ob2.load_state_dict(saved)
    run(b2, ob2, xs[3:])
    oc = torch.optim.AdamW(c.parameters(), lr=3e-3, weight_decay=1e-2)
    run(c, oc, xs)
    for (k, pa), (_, pb), (_, pc) in zip(a.named_parameters(), b2.named_parameters(), c.named_parameters()):
        assert torch.equal(pa, pb), k                       # resume is exact
        assert rel_l2(pa, pc) < 2e-6, k
    assert torch.equal(a.unused, b2.unused)


def test_first_step_bias_correction_matches_torch_double_arithmetic():
    """step 1: 1 - beta^1 evaluated in double on the host.  lr = 1 makes the update (+-1 per element at step 1)
    dominate the parameter, so an error in lr / bias1 shows up undiluted."""
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    a, b = _toy(), _toy()
    oa = torch.optim.AdamW(a.parameters(), lr=1.0, betas=(0.95, 0.999), weight_decay=0.0)
    ob = FusedAdamW(b.parameters(), lr=1.0, betas=(0.95, 0.999), weight_decay=0.0)
    x = torch.randn(16, 37, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    for m, o in ((a, oa), (b, ob)):
        o.zero_grad()
        m(x).square().sum().backward()
        o.step()
    for (k, pa), (_, pb) in zip(a.named_parameters(), b.named_parameters()):
        if k == "unused":
            continue
        assert rel_l2(pb, pa) < 5e-7, k


def test_rel_l2_backward_is_finite_for_zero_weight_and_exact_prediction():
    """reduction=False with a zero per-sample weight, and a sample with pred == y (dnorm = 0): the reference's
    torch.norm backward gives 0 there; the fused kernel must not put inf/NaN into the gradient bucket."""
    from transformerbasednavierstokesolver_amd.utils.testloss import TestLoss, FusedTestLoss
    g = torch.Generator(device=DEV).manual_seed(3)
    y = torch.randn(4, 300, device=DEV, generator=g)
    pred = torch.randn(4, 300, device=DEV, generator=g)
    pred[2] = y[2]                                           # exact prediction
    w = torch.tensor([1.0, 0.0, 2.0, 0.5], device=DEV)       # masked sample
    p1 = pred.clone().requires_grad_(True)
    p2 = pred.clone().requires_grad_(True)
    (FusedTestLoss(reduction=False)(p1, y) * w).sum().backward()
    (TestLoss(reduction=False)(p2, y) * w).sum().backward()
    assert torch.isfinite(p1.grad).all()
    assert float(p1.grad[1].abs().max()) == 0.0 and float(p1.grad[2].abs().max()) == 0.0
    assert torch.isfinite(p2.grad).all() and rel_l2(p1.grad, p2.grad) < 5e-6


def test_rollout_graph_captured_before_the_optimizer_refuses_to_replay_stale_parameters():
    """FusedAdamW re-seats every parameter into its flat buffer at construction.  A GraphedRollout captured BEFORE
    that holds the old pointers: run() must raise instead of replaying stale weights; captured AFTER, it follows
    the optimizer's updates."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    from transformerbasednavierstokesolver_amd.utils.testloss import FusedTestLoss
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    m = harness.build_model(cfg, synth.synth_state_dict(cfg, seed=191), DEV)
    pos, a, u = synth.ns_batch(2, seed=192)
    x, fx, yy = (torch.from_numpy(t).to(DEV) for t in (pos, a, u[..., :2]))
    early = harness.GraphedRollout(m.eval(), x, fx)
    early.run(fx, 1)
    opt = FusedAdamW(m.parameters(), lr=1e-2, weight_decay=1e-5)
    with pytest.raises(RuntimeError, match="parameter storage moved"):
        early.run(fx, 1)
    late = harness.GraphedRollout(m.eval(), x, fx)
    before = late.run(fx, 2)
    harness.train_step(m.train(), opt, None, x, fx, yy, grad_sync=opt.sync, loss_fn=FusedTestLoss(size_average=False))
    after = late.run(fx, 2)
    assert torch.equal(after, harness.rollout(m.eval(), x, fx, 2))      # the graph sees the updated weights
    assert not torch.equal(after, before)


def test_parameter_first_used_after_the_optimizer_was_built_is_stepped():
    """`placeholder` gets no gradient while fx is given (…_2D.py:205-210) and must stay bit-identical (no weight
    decay); once a call with fx=None uses it, it is reduced and stepped like torch.optim.AdamW would."""
    from transformerbasednavierstokesolver_amd import synth, harness
    from transformerbasednavierstokesolver_amd.optim import FusedAdamW
    cfg = dict(synth.TINY_CONFIG, fun_dim=0, unified_pos=0)
    sd = synth.synth_state_dict(cfg, seed=201)
    N = cfg["H"] * cfg["W"]
    rng = np.random.default_rng(202)
    x = torch.from_numpy(rng.standard_normal((2, N, 2)).astype(np.float32)).to(DEV)
    m = harness.build_model(cfg, sd, DEV).train()
    opt = FusedAdamW(m.parameters(), lr=1e-2, weight_decay=1e-1)
    p0 = m.placeholder.detach().clone()
    # step 1: only the head parameters get gradients (loss on a detached trunk): placeholder untouched
    opt.zero_grad()
    m.blocks[-1].mlp2.weight.square().sum().backward()
    opt.sync()
    opt.step()
    assert torch.equal(m.placeholder.detach(), p0) and m.placeholder.grad is None
    # step 2: fx=None -> placeholder is used
    opt.zero_grad()
    m(x, None).square().sum().backward()
    opt.sync()
    opt.step()
    assert not torch.equal(m.placeholder.detach(), p0)
    assert m.placeholder.grad is not None and m.placeholder.grad.data_ptr() == opt.sync.views[opt.sync.index_of(m.placeholder)].data_ptr()
