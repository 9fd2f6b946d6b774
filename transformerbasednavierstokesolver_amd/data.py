"""Data ingress either side of the hot path.

* `load_ns_mat` / `split_ns_trajectories`   the `.mat` -> (a, u) slicing of exp_ns.py:64-80
* `load_darcy_mat` / `split_darcy`           exp_darcy.py:76-99 (coeff/sol fields + UnitTransformer encoding)
* `grid_positions`                          the driver-side position input (exp_ns.py:88-94)
* `ResidentDataset`                         the whole split kept in HBM and batched by device-side index
                                            selection: replaces TensorDataset + DataLoader(num_workers=0) +
                                            per-batch `.cuda()` (exp_ns.py:96-99,195); one NS split at 64x64
                                            is 1200 x 4096 x 20 x 4 B = 393 MB, nothing against 288 GB
* `simulate_ns_vorticity`                   torch.fft pseudo-spectral 2-D Navier-Stokes (vorticity form,
                                            FNO forcing) standing in for the PhiFlow data-generation notebook.
                                            PARITY UNPINNED: the notebook's generator (PhiFlow + an external
                                            repo) cannot run here; this is a utility, tested on invariants only.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from .utils.normalizer import UnitTransformer


def _downsampled(field, r, h):
    return field[:, ::r, ::r][:, :h, :h]


def split_ns_trajectories(u, ntrain, ntest, T_in=10, T=10, r=1):
    """u: [S, 64, 64, >=T_in+T].  First `ntrain` trajectories train, LAST `ntest` test; inputs are frames
    [0, T_in), targets frames [T_in, T_in+T); spatial stride r; points flattened row-major."""
    u = np.asarray(u)
    h = int(((u.shape[1] - 1) / r) + 1)

    def cut(block, t0, t1):
        x = _downsampled(block[..., t0:t1], r, h)
        return torch.from_numpy(np.ascontiguousarray(x.reshape(x.shape[0], -1, x.shape[-1])))

    tr, te = u[:ntrain], u[-ntest:]
    return dict(h=h, train_a=cut(tr, 0, T_in), train_u=cut(tr, T_in, T_in + T),
                test_a=cut(te, 0, T_in), test_u=cut(te, T_in, T_in + T))


def load_ns_mat(path, ntrain, ntest, T_in=10, T=10, r=1, key="u"):
    import scipy.io as scio
    return split_ns_trajectories(scio.loadmat(path)[key], ntrain, ntest, T_in, T, r)


def split_darcy(coeff, sol, n, r=1):
    """[S, 421, 421] fields -> ([n, s*s] float32 coefficient, [n, s*s] solution, s)"""
    s = int(((coeff.shape[1] - 1) / r) + 1)
    x = torch.from_numpy(np.ascontiguousarray(_downsampled(np.asarray(coeff)[:n], r, s).reshape(n, -1))).float()
    y = torch.from_numpy(np.ascontiguousarray(_downsampled(np.asarray(sol)[:n], r, s).reshape(n, -1)))
    return x, y, s


def load_darcy_mat(train_path, test_path, ntrain, ntest, r=1):
    """Returns encoded train/test tensors, the two normalisers (fitted on the TRAIN split), s and dx = 1/s."""
    import scipy.io as scio
    tr, te = scio.loadmat(train_path), scio.loadmat(test_path)
    x_train, y_train, s = split_darcy(tr["coeff"], tr["sol"], ntrain, r)
    x_test, y_test, _ = split_darcy(te["coeff"], te["sol"], ntest, r)
    xn, yn = UnitTransformer(x_train), UnitTransformer(y_train)
    return dict(s=s, dx=1.0 / s, x_normalizer=xn, y_normalizer=yn, x_train=xn.encode(x_train),
                y_train=yn.encode(y_train), x_test=xn.encode(x_test), y_test=y_test)


def grid_positions(h, w=None):
    """[1, h*w, 2] float32; 'xy' meshgrid, i.e. the FIRST coordinate varies along image columns."""
    w = h if w is None else w
    xx, yy = np.meshgrid(np.linspace(0, 1, w), np.linspace(0, 1, h))
    return torch.tensor(np.c_[xx.ravel(), yy.ravel()], dtype=torch.float).unsqueeze(0)


class ResidentDataset:
    """Tensors with a common leading sample dimension, resident on one device."""

    def __init__(self, *tensors, device=None):
        n = tensors[0].shape[0]
        if any(t.shape[0] != n for t in tensors):
            raise ValueError("all tensors need the same number of samples")
        self.tensors = [t.to(device).contiguous() if device is not None else t.contiguous() for t in tensors]

    def __len__(self):
        return self.tensors[0].shape[0]

    def shard(self, rank, world_size):
        """contiguous equal shards (drops the remainder, like DistributedSampler(drop_last=True))"""
        n = len(self) // world_size
        return ResidentDataset(*[t[rank * n:(rank + 1) * n] for t in self.tensors])

    def batches(self, batch_size, shuffle=False, generator=None, drop_last=False):
        n, dev = len(self), self.tensors[0].device
        order = torch.randperm(n, device=dev, generator=generator) if shuffle else torch.arange(n, device=dev)
        for i in range(0, n, batch_size):
            idx = order[i:i + batch_size]
            if drop_last and idx.numel() < batch_size:
                return
            yield tuple(t.index_select(0, idx) for t in self.tensors)


@torch.no_grad()
def simulate_ns_vorticity(w0, visc=1e-5, T=20.0, dt=1e-3, record=20, forcing=True):
    """Pseudo-spectral 2-D Navier-Stokes in vorticity form on the periodic unit square:
    w_t + u.grad(w) = visc * lap(w) + f,  f = 0.1 (sin 2pi(x+y) + cos 2pi(x+y)),  Crank-Nicolson for the
    viscous term, explicit advection, 2/3 de-aliasing.  w0: [S, n, n].  Returns [S, n, n, record]."""
    S, n, _ = w0.shape
    dev, dt64 = w0.device, torch.float64
    k = torch.fft.fftfreq(n, d=1.0 / n, device=dev).to(dt64)
    ky, kx = k.reshape(n, 1).expand(n, n), k.reshape(1, n).expand(n, n)
    lap = 4 * math.pi ** 2 * (kx ** 2 + ky ** 2)
    lap_safe = lap.clone()
    lap_safe[0, 0] = 1.0
    dealias = ((kx.abs() <= n / 3) & (ky.abs() <= n / 3)).to(dt64)
    grid = torch.linspace(0, 1, n + 1, device=dev, dtype=dt64)[:-1]
    xs, ys = grid.reshape(1, n), grid.reshape(n, 1)
    f_h = torch.fft.fft2(0.1 * (torch.sin(2 * math.pi * (xs + ys)) + torch.cos(2 * math.pi * (xs + ys)))) if forcing else 0.0
    w_h = torch.fft.fft2(w0.to(dt64))
    steps = int(round(T / dt))
    every = max(1, steps // record)
    out = torch.empty(S, n, n, record, dtype=torch.float32, device=dev)
    rec = 0
    for it in range(steps):
        psi_h = w_h / lap_safe
        u = torch.fft.ifft2(2j * math.pi * ky * psi_h).real          # u =  d psi / dy
        v = torch.fft.ifft2(-2j * math.pi * kx * psi_h).real         # v = -d psi / dx
        wx = torch.fft.ifft2(2j * math.pi * kx * w_h).real
        wy = torch.fft.ifft2(2j * math.pi * ky * w_h).real
        adv_h = torch.fft.fft2(u * wx + v * wy) * dealias
        w_h = (w_h * (1.0 - 0.5 * dt * visc * lap) + dt * (f_h - adv_h)) / (1.0 + 0.5 * dt * visc * lap)
        if (it + 1) % every == 0 and rec < record:
            out[..., rec] = torch.fft.ifft2(w_h).real.float()
            rec += 1
    return out
