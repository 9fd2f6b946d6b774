"""SURVEY §8(f)-4 on the device: the resident split batches / shards by device-side index selection, feeds the HIP
training step without a host round trip (exp_ns.py:96-99,195 kept TensorDataset + DataLoader + per-batch .cuda()),
and the torch.fft Navier-Stokes generator gives the same fields on the GPU as on the CPU."""
import numpy as np
import pytest
import torch

from conftest import rel_l2
from transformerbasednavierstokesolver_amd import data, synth, harness

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_resident_dataset_on_device_batches_shards_and_feeds_the_training_step():
    pos, a, u = synth.ns_batch(6, seed=301)
    split = data.split_ns_trajectories(np.concatenate((a, u), -1).reshape(6, 64, 64, 20), ntrain=4, ntest=2)
    assert torch.equal(split["train_a"], torch.from_numpy(a[:4])) and torch.equal(split["test_u"], torch.from_numpy(u[-2:]))
    x = data.grid_positions(64).repeat(4, 1, 1)
    ds = data.ResidentDataset(x, split["train_a"], split["train_u"], device=DEV)
    assert all(t.is_cuda for t in ds.tensors) and len(ds) == 4
    g = torch.Generator(device=DEV).manual_seed(5)
    seen = []
    for bx, ba, bu in ds.batches(3, shuffle=True, generator=g):
        assert bx.is_cuda and ba.is_cuda and bu.is_cuda and ba.is_contiguous()
        seen.append(ba)
    seen = torch.cat(seen)
    assert seen.shape[0] == 4
    # every trajectory exactly once, bit-identical to the host copy
    host = split["train_a"]
    match = (seen.cpu()[:, None] == host[None]).flatten(2).all(-1)
    assert torch.equal(match.sum(0), torch.ones(4, dtype=torch.long)) and torch.equal(match.sum(1), torch.ones(4, dtype=torch.long))
    # shards are the contiguous per-rank ranges of ddp.shard_batch and stay on the device
    sh0, sh1 = ds.shard(0, 2), ds.shard(1, 2)
    assert torch.equal(sh0.tensors[1], ds.tensors[1][:2]) and torch.equal(sh1.tensors[1], ds.tensors[1][2:])
    assert sh1.tensors[1].is_cuda
    # a device-side batch drives the HIP training step directly and equals the step fed from host tensors
    cfg = dict(synth.NS_SMALL_CONFIG, n_layers=2)
    sd = synth.synth_state_dict(cfg, seed=302)
    losses = []
    for src in ("resident", "host"):
        m = harness.build_model(cfg, sd, DEV).train()
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5)
        if src == "resident":
            bx, ba, bu = next(iter(ds.batches(2)))
        else:
            bx, ba, bu = x[:2].to(DEV), host[:2].to(DEV), split["train_u"][:2].to(DEV)
        loss, _ = harness.train_step(m, opt, None, bx, ba, bu[..., :2])
        losses.append((float(loss), {k: p.detach().clone() for k, p in m.named_parameters()}))
    assert losses[0][0] == losses[1][0]
    for k in losses[0][1]:
        assert torch.equal(losses[0][1][k], losses[1][1][k]), k


def test_ns_generator_on_the_gpu_matches_the_cpu_run():
    """simulate_ns_vorticity integrates in float64 with torch.fft on whatever device holds w0: the GPU run must
    reproduce the CPU run (different FFT libraries: rocFFT vs pocketfft -> agreement to ~1e-10, not bitwise)."""
    w0 = torch.from_numpy(synth.synth_ns_fields(3, 64, 64, 1, seed=7)[..., 0])
    kw = dict(visc=1e-3, T=0.1, dt=1e-3, record=5, forcing=True)
    cpu = data.simulate_ns_vorticity(w0, **kw)
    gpu = data.simulate_ns_vorticity(w0.to(DEV), **kw)
    assert gpu.is_cuda and gpu.shape == (3, 64, 64, 5) and torch.isfinite(gpu).all()
    assert rel_l2(gpu, cpu) < 1e-6                      # outputs are stored as float32
    # and the generated frames are a valid exp_ns input: split + one model call on the device
    sp = data.split_ns_trajectories(gpu.cpu().numpy(), ntrain=2, ntest=1, T_in=3, T=2)
    assert sp["train_a"].shape == (2, 4096, 3) and sp["test_u"].shape == (1, 4096, 2)


def test_resident_split_on_the_device_equals_the_oracle_restatement():
    """The device-resident split (what the training loop actually reads) against oracle/data_oracle.py — the restatement of
    exp_ns.py:61-94 and exp_darcy.py:71-96 — not against the host copy of the same product code: bit-exact for the slicing
    and the positions, fp32 rounding for the UnitTransformer encode evaluated on the GPU."""
    from oracle import data_oracle as dorc
    from transformerbasednavierstokesolver_amd.utils.normalizer import UnitTransformer
    u = synth.synth_ns_fields(8, 64, 64, 20, seed=21)
    for r in (1, 2):
        want = dorc.ns_split(u, 5, 2, 10, 10, r)
        sp = data.split_ns_trajectories(u, ntrain=5, ntest=2, T_in=10, T=10, r=r)
        ds = data.ResidentDataset(data.grid_positions(sp["h"]).repeat(5, 1, 1), sp["train_a"], sp["train_u"], device=DEV)
        got = [t.cpu().numpy() for t in next(iter(ds.batches(5)))]
        assert np.array_equal(got[0], dorc.ns_positions(want["h"], 5))
        assert np.array_equal(got[1], want["train_a"]) and np.array_equal(got[2], want["train_u"])
        te = data.ResidentDataset(sp["test_a"], sp["test_u"], device=DEV)
        assert np.array_equal(te.tensors[0].cpu().numpy(), want["test_a"]) and np.array_equal(te.tensors[1].cpu().numpy(), want["test_u"])
    rng = np.random.default_rng(9)
    coeff, sol = rng.choice([3.0, 12.0], size=(3, 421, 421)), rng.standard_normal((3, 421, 421)) * 0.01
    x, y, s = data.split_darcy(coeff, sol, 3, 5)
    xo, yo, so, _ = dorc.darcy_split(coeff, sol, 3, 5)
    assert s == so
    xd, yd = x.to(DEV), y.to(DEV)
    xn, yn = UnitTransformer(xd), UnitTransformer(yd)              # fitted and applied on the device
    mx, sx = dorc.unit_fit(xo.astype(np.float64))
    my, sy = dorc.unit_fit(yo)
    assert rel_l2(xn.encode(xd), dorc.unit_encode(xo.astype(np.float64), mx, sx)) < 1e-5
    assert rel_l2(yn.encode(yd), dorc.unit_encode(yo, my, sy)) < 1e-12
    assert rel_l2(yn.decode(yn.encode(yd)), yo) < 1e-12
