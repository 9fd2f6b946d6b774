"""Sinusoidal timestep embedding (reference: model/Embedding.py:67-85).  Host-side glue: it acts on
a [B,1] tensor once per call and is only used when `Time_Input=True` (not on the NS/Darcy path)."""
import math

import torch


def timestep_embedding(timesteps, dim, max_period=10000, repeat_only=False):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    args = timesteps[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[..., :1])], dim=-1)
    return emb
