"""Build-owned deterministic generators: model weights and Navier-Stokes-like fields.

The reference ships no datasets (its `data/` is git-ignored) and does not seed anything, so
parity is defined on identical weights + inputs.  Everything here is a pure
function of (config, seed) through `numpy.random.default_rng`, so the GPU box can regenerate the
exact tensors the golden fixtures were captured on without access to the reference.

State-dict layout follows the reference contract (key names / shapes of
model/Transolver_Structured_Mesh_2D.py:122-181 and model/Physics_Attention.py:62-86).
"""
from __future__ import annotations

import numpy as np

__all__ = ["darcy_batch", "DARCY_CONFIG", "make_config", "state_dict_spec", "synth_state_dict", "synth_ns_fields", "ns_batch",
           "meshgrid_pos", "NS_CONFIG", "NS_SMALL_CONFIG", "TINY_CONFIG"]


def make_config(space_dim=2, n_layers=8, n_hidden=256, n_head=8, mlp_ratio=1, fun_dim=10, out_dim=1,
                slice_num=64, ref=8, unified_pos=1, H=64, W=64, Time_Input=False, act="gelu",
                dropout=0.0):
    return dict(space_dim=space_dim, n_layers=n_layers, n_hidden=n_hidden, n_head=n_head,
                mlp_ratio=mlp_ratio, fun_dim=fun_dim, out_dim=out_dim, slice_num=slice_num, ref=ref,
                unified_pos=int(unified_pos), H=H, W=W, Time_Input=bool(Time_Input), act=act,
                dropout=dropout)


# BASELINE.json configs[1]: NS 64x64, 8 layers, C=256, 8 heads, M=64 slices.
NS_CONFIG = make_config()
# Size of the checkpoints the reference ships (checkpoints/ep400_sim100.pt): C=64, M=32.
NS_SMALL_CONFIG = make_config(n_hidden=64, slice_num=32)
# Tiny parity case (non-square grid, D=8, M=12 not a multiple of 16, out_dim 2).
TINY_CONFIG = make_config(n_layers=2, n_hidden=32, n_head=4, fun_dim=3, out_dim=2, slice_num=12,
                          ref=3, unified_pos=1, H=6, W=5, mlp_ratio=2)


def in_features(cfg):
    return cfg["fun_dim"] + (cfg["ref"] ** 2 if cfg["unified_pos"] else cfg["space_dim"])


def state_dict_spec(cfg):
    """Ordered (key, shape, kind) triples of the reference Model.state_dict()."""
    C, L, h = cfg["n_hidden"], cfg["n_layers"], cfg["n_head"]
    D, M, r = C // h, cfg["slice_num"], cfg["mlp_ratio"]
    spec = [("placeholder", (C,), "placeholder"),
            ("preprocess.linear_pre.0.weight", (2 * C, in_features(cfg)), "lin_w"),
            ("preprocess.linear_pre.0.bias", (2 * C,), "bias"),
            ("preprocess.linear_post.weight", (C, 2 * C), "lin_w"),
            ("preprocess.linear_post.bias", (C,), "bias")]
    if cfg["Time_Input"]:
        spec += [("time_fc.0.weight", (C, C), "lin_w"), ("time_fc.0.bias", (C,), "bias"),
                 ("time_fc.2.weight", (C, C), "lin_w"), ("time_fc.2.bias", (C,), "bias")]
    for i in range(L):
        p = f"blocks.{i}."
        spec += [(p + "ln_1.weight", (C,), "ln_w"), (p + "ln_1.bias", (C,), "bias"),
                 (p + "Attn.temperature", (1, h, 1, 1), "temperature"),
                 (p + "Attn.in_project_x.weight", (C, C, 3, 3), "conv_w"),
                 (p + "Attn.in_project_x.bias", (C,), "bias"),
                 (p + "Attn.in_project_fx.weight", (C, C, 3, 3), "conv_w"),
                 (p + "Attn.in_project_fx.bias", (C,), "bias"),
                 (p + "Attn.in_project_slice.weight", (M, D), "slice_w"),
                 (p + "Attn.in_project_slice.bias", (M,), "bias"),
                 (p + "Attn.to_q.weight", (D, D), "qk_w"),
                 (p + "Attn.to_k.weight", (D, D), "qk_w"),
                 (p + "Attn.to_v.weight", (D, D), "lin_w"),
                 (p + "Attn.to_out.0.weight", (C, C), "lin_w"),
                 (p + "Attn.to_out.0.bias", (C,), "bias"),
                 (p + "ln_2.weight", (C,), "ln_w"), (p + "ln_2.bias", (C,), "bias"),
                 (p + "mlp.linear_pre.0.weight", (r * C, C), "lin_w"),
                 (p + "mlp.linear_pre.0.bias", (r * C,), "bias"),
                 (p + "mlp.linear_post.weight", (C, r * C), "lin_w"),
                 (p + "mlp.linear_post.bias", (C,), "bias")]
        if i == L - 1:
            spec += [(p + "ln_3.weight", (C,), "ln_w"), (p + "ln_3.bias", (C,), "bias"),
                     (p + "mlp2.weight", (cfg["out_dim"], C), "lin_w"),
                     (p + "mlp2.bias", (cfg["out_dim"],), "bias")]
    return spec


def irregular_state_dict_spec(cfg):
    """state_dict of model/Transolver_Irregular_Mesh.Model: same keys, Linear [C,C] projections."""
    out = []
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "conv_w":
            out.append((key, shape[:2], "lin_w"))
        else:
            out.append((key, shape, kind))
    return out


def synth_irregular_state_dict(cfg, seed=0):
    rng = np.random.default_rng(seed)
    sd = {}
    for key, shape, kind in irregular_state_dict_spec(cfg):
        if kind in ("lin_w",):
            v = rng.standard_normal(shape) / np.sqrt(shape[1])
        elif kind == "qk_w":
            v = rng.standard_normal(shape) * (1.5 / np.sqrt(shape[1]))
        elif kind == "slice_w":
            v = rng.standard_normal(shape) / np.sqrt(shape[1])
        elif kind == "bias":
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "ln_w":
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "temperature":          # NOT clamped in this family: include values outside [0.1, 5]
            v = np.resize(np.array([0.06, 0.5, 6.0, 0.25, 1.5, 0.8, 0.3, 1.1]), shape[1]).reshape(shape)
        else:
            v = rng.uniform(0.0, 1.0, size=shape) / shape[0]
        sd[key] = np.ascontiguousarray(v, dtype=np.float32)
    return sd


def synth_state_dict(cfg, seed=0, wild_temperature=False):
    """Deterministic fp32 weights (numpy) with O(1) activations through the whole stack.

    Scales are chosen so that the slice softmax and the token attention are far from uniform
    (the reference's trunc_normal(0.02) init leaves to_q/to_k gradients ~1e-9), which
    makes parity tests discriminating.  `wild_temperature` puts some heads outside [0.1, 5] to
    pin the clamp mask of Physics_Attention.py:98-99.
    """
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape, kind in state_dict_spec(cfg):
        if kind == "lin_w":
            v = rng.standard_normal(shape) / np.sqrt(shape[1])
        elif kind == "qk_w":
            v = rng.standard_normal(shape) * (1.5 / np.sqrt(shape[1]))
        elif kind == "slice_w":
            v = rng.standard_normal(shape) * (1.0 / np.sqrt(shape[1]))
        elif kind == "conv_w":
            v = rng.standard_normal(shape) / np.sqrt(9.0 * shape[1])
        elif kind == "bias":
            v = 0.1 * rng.standard_normal(shape)
        elif kind == "ln_w":
            v = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "temperature":
            if wild_temperature:
                pool = np.array([0.03, 0.5, 7.0, 0.25, 1.5, 0.1, 5.0, 0.8])
                v = np.resize(pool, shape[1]).reshape(shape)
            else:
                v = rng.uniform(0.3, 1.2, size=shape)
        elif kind == "placeholder":
            v = rng.uniform(0.0, 1.0, size=shape) / shape[0]
        else:  # pragma: no cover
            raise KeyError(kind)
        out[key] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def synth_ns_fields(S, H=64, W=64, T=20, seed=0):
    """[S,H,W,T] fp32 vorticity-like trajectories: low-passed Gaussian noise whose
    Fourier modes rotate (advection) and decay slowly from frame to frame; unit std overall."""
    rng = np.random.default_rng(seed)
    ky = np.fft.fftfreq(H, d=1.0 / H)[:, None]
    kx = np.fft.fftfreq(W, d=1.0 / W)[None, :]
    k2 = ky ** 2 + kx ** 2
    amp = (1.0 + k2 / 16.0) ** -1.25
    out = np.empty((S, H, W, T), dtype=np.float32)
    for s in range(S):
        spec = np.fft.fft2(rng.standard_normal((H, W))) * amp
        vel = rng.uniform(-1.5, 1.5, size=2)
        phase = np.exp(-2j * np.pi * (ky * vel[0] / H + kx * vel[1] / W))
        decay = np.exp(-1e-3 * k2)
        for t in range(T):
            out[s, :, :, t] = np.real(np.fft.ifft2(spec))
            spec = spec * phase * decay
    out /= out.std()
    return out


def meshgrid_pos(S, H, W):
    """Driver-side plain position input, exp_ns.py:88-94 convention ('xy' meshgrid: the first
    coordinate varies along image columns)."""
    x = np.linspace(0, 1, W)
    y = np.linspace(0, 1, H)
    xx, yy = np.meshgrid(x, y)
    pos = np.stack([xx.ravel(), yy.ravel()], axis=-1).astype(np.float32)
    return np.array(np.broadcast_to(pos[None], (S, H * W, 2)))   # writable copy


def ns_batch(S, H=64, W=64, T_in=10, T=10, seed=0):
    """(pos [S,N,2], a [S,N,T_in], u [S,N,T]) with the exp_ns.py:64-80 split of a trajectory."""
    f = synth_ns_fields(S, H, W, T_in + T, seed)
    a = np.ascontiguousarray(f[..., :T_in].reshape(S, H * W, T_in))
    u = np.ascontiguousarray(f[..., T_in:].reshape(S, H * W, T))
    return meshgrid_pos(S, H, W), a, u


# BASELINE.json configs[4]: Darcy 421x421 structured mesh, 8 layers, C=128, 8 heads, M=128 slices
# (exp_darcy.py:118-130 constructor arguments: fun_dim=1, out_dim=1, unified_pos + ref 8 as scripts/Transolver_Darcy.sh).
DARCY_CONFIG = make_config(n_layers=8, n_hidden=128, n_head=8, slice_num=128, fun_dim=1, out_dim=1, H=421, W=421)


def darcy_batch(S, s=421, seed=0):
    """(pos [S,N,2], coeff [S,N], sol [S,N]) shaped like exp_darcy.py:76-83: `coeff` a piecewise-constant {3, 12}
    permeability field (thresholded low-passed noise, as the FNO Darcy set), `sol` a smooth O(1e-2) field standing
    in for the pressure solution (values only need the right shape and smoothness; throughput is value-independent)."""
    rng = np.random.default_rng(seed)
    k = np.fft.fftfreq(s, d=1.0 / s)
    k2 = k[:, None] ** 2 + k[None, :] ** 2
    coeff = np.empty((S, s * s), dtype=np.float32)
    sol = np.empty((S, s * s), dtype=np.float32)
    for i in range(S):
        g = np.real(np.fft.ifft2(np.fft.fft2(rng.standard_normal((s, s))) * (1.0 + k2 / 9.0) ** -1.0))
        coeff[i] = np.where(g > 0, 12.0, 3.0).ravel()
        u = np.real(np.fft.ifft2(np.fft.fft2(rng.standard_normal((s, s))) * (1.0 + k2 / 4.0) ** -1.5))
        sol[i] = (0.01 * u / u.std() + 0.02).ravel()
    return meshgrid_pos(S, s, s), coeff, sol
