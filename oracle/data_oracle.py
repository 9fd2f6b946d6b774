"""TEST INFRASTRUCTURE — CPU restatement of the reference's data ingress either side of the hot path.

Only tests/ may import this file.  Plain numpy, written from the reference text one statement at a time (no shared
helper with the product's data.py, so that a slip in either shows):
  ns_split        exp_ns.py:61-80      `.mat` array `u` [S, 64, 64, T] -> train/test (a, u) point clouds
  ns_positions    exp_ns.py:88-94      np.meshgrid('xy') positions, float32, repeated per sample
  darcy_split     exp_darcy.py:71-89   coeff / sol [S, 421, 421] -> [n, s*s] (coefficient cast to float32)
  unit_encode     utils/normalizer.py:30-52 + exp_darcy.py:91-96   mean / std over dims (0, 1), keepdim, +1e-8
PARITY: pinned by construction only for what numpy defines (strided slicing, C-order reshape, meshgrid): the reference
drivers execute argparse and open a hard-coded Windows path at import, so they cannot be imported here, and the datasets
(`data/`, git-ignored) are absent — the arrays below are synthetic of the same shapes.
"""
import numpy as np


def ns_split(u, ntrain, ntest, T_in, T, r):
    # exp_ns.py:62  h = int(((64 - 1) / r) + 1)
    h = int(((64 - 1) / r) + 1)
    out = {"h": h}
    # exp_ns.py:67-69 / 71-73: data['u'][:ntrain, ::r, ::r, :T_in][:, :h, :h, :] ; reshape(S, -1, T)
    a = u[:ntrain, ::r, ::r, :T_in][:, :h, :h, :]
    out["train_a"] = a.reshape(a.shape[0], -1, a.shape[-1])
    b = u[:ntrain, ::r, ::r, T_in:T + T_in][:, :h, :h, :]
    out["train_u"] = b.reshape(b.shape[0], -1, b.shape[-1])
    # exp_ns.py:75-80: the LAST ntest trajectories
    a = u[-ntest:, ::r, ::r, :T_in][:, :h, :h, :]
    out["test_a"] = a.reshape(a.shape[0], -1, a.shape[-1])
    b = u[-ntest:, ::r, ::r, T_in:T + T_in][:, :h, :h, :]
    out["test_u"] = b.reshape(b.shape[0], -1, b.shape[-1])
    return out


def ns_positions(h, n):
    # exp_ns.py:88-94
    x = np.linspace(0, 1, h)
    y = np.linspace(0, 1, h)
    x, y = np.meshgrid(x, y)
    pos = np.c_[x.ravel(), y.ravel()]
    pos = pos.astype(np.float32)[None]               # torch.tensor(pos, dtype=torch.float).unsqueeze(0)
    return np.repeat(pos, n, axis=0)                 # pos.repeat(n, 1, 1)


def darcy_split(coeff, sol, n, r):
    # exp_darcy.py:72-74
    h = int(((421 - 1) / r) + 1)
    s = h
    # exp_darcy.py:77-82
    x = coeff[:n, ::r, ::r][:, :s, :s]
    x = x.reshape(n, -1).astype(np.float32)          # torch.from_numpy(x_train).float()
    y = sol[:n, ::r, ::r][:, :s, :s]
    y = y.reshape(n, -1)
    return x, y, s, 1.0 / s


def unit_fit(X):
    # utils/normalizer.py:32-33: mean / std over dims (0, 1) keepdim, std + 1e-8 (torch.std is the unbiased estimator)
    mean = X.mean(axis=(0, 1), keepdims=True)
    std = X.std(axis=(0, 1), keepdims=True, ddof=1) + 1e-8
    return mean, std


def unit_encode(x, mean, std):
    return (x - mean) / std                          # utils/normalizer.py:44


def unit_decode(x, mean, std):
    return x * std + mean                            # utils/normalizer.py:47-48
