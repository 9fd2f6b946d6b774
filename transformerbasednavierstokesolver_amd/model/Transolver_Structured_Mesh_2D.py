"""Transolver for structured 2-D meshes — drop-in for the reference module of the same name
(model/Transolver_Structured_Mesh_2D.py): identical `Model(...)` keyword signature (:122-139),
`forward(x, fx, T=None)` (:202-220), `MLP` (:13-38), `Transolver_block` (:41-75) and the
169-tensor state_dict layout, computed by the libpa2d HIP kernels.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn
from .Embedding import timestep_embedding
from .Physics_Attention import Physics_Attention_Structured_Mesh_2D

# names of the reference's ACTIVATION table (:9-10); values are only used as parameter-free markers
ACTIVATION = {'gelu': nn.GELU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid, 'relu': nn.ReLU,
              'softplus': nn.Softplus, 'ELU': nn.ELU, 'silu': nn.SiLU}


def _pad4(x, w):
    """libpa2d GEMMs need the contraction length to be a multiple of 4 (16-byte loads)."""
    k = w.shape[1]
    if k % 4:
        pad = 4 - k % 4
        x = F.pad(x, (0, pad))
        w = F.pad(w, (0, pad))
    return x, w


class MLP(nn.Module):
    def __init__(self, n_input, n_hidden, n_output, n_layers=1, act='gelu', res=True):
        super(MLP, self).__init__()
        if act not in ACTIVATION:
            # (the reference's 'leaky_relu' entry is an instance, not a class, and fails at act())
            raise NotImplementedError
        self.act_name = act
        self.n_input = n_input
        self.n_hidden = n_hidden
        self.n_output = n_output
        self.n_layers = n_layers
        self.res = res
        self.linear_pre = nn.Sequential(nn.Linear(n_input, n_hidden), ACTIVATION[act]())
        self.linear_post = nn.Linear(n_hidden, n_output)
        self.linears = nn.ModuleList([nn.Sequential(nn.Linear(n_hidden, n_hidden), ACTIVATION[act]())
                                      for _ in range(n_layers)])

    def forward(self, x, residual=None):
        lin = self.linear_pre[0]
        x, w1 = _pad4(x, lin.weight)
        if self.n_layers == 0:
            return Fn.mlp(x, residual, self.act_name, w1, lin.bias, self.linear_post.weight, self.linear_post.bias)
        h = Fn.linear(x, w1, lin.bias, self.act_name)
        for seq in self.linears:
            y = Fn.linear(h, seq[0].weight, seq[0].bias, self.act_name)
            h = y + h if self.res else y
        out = Fn.linear(h, self.linear_post.weight, self.linear_post.bias, None)
        return out if residual is None else out + residual


class Transolver_block(nn.Module):
    """Transformer encoder block."""

    def __init__(self, num_heads, hidden_dim, dropout, act='gelu', mlp_ratio=4, last_layer=False, out_dim=1,
                 slice_num=32, H=85, W=85):
        super().__init__()
        self.last_layer = last_layer
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.Attn = Physics_Attention_Structured_Mesh_2D(hidden_dim, heads=num_heads,
                                                         dim_head=hidden_dim // num_heads, dropout=dropout,
                                                         slice_num=slice_num, H=H, W=W)
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.mlp = MLP(hidden_dim, hidden_dim * mlp_ratio, hidden_dim, n_layers=0, res=False, act=act)
        if self.last_layer:
            self.ln_3 = nn.LayerNorm(hidden_dim)
            self.mlp2 = nn.Linear(hidden_dim, out_dim)

    def forward(self, fx):
        fx = self.Attn(Fn.layer_norm(fx, self.ln_1.weight, self.ln_1.bias), residual=fx)
        fx = self.mlp(Fn.layer_norm(fx, self.ln_2.weight, self.ln_2.bias), residual=fx)
        if self.last_layer:
            z = Fn.layer_norm(fx, self.ln_3.weight, self.ln_3.bias)
            if self.mlp2.out_features <= 8:
                return Fn.head(z, self.mlp2.weight, self.mlp2.bias)
            return Fn.linear(z, self.mlp2.weight, self.mlp2.bias, None)
        return fx


class Model(nn.Module):
    def __init__(self,
                 space_dim=1,
                 n_layers=5,
                 n_hidden=256,
                 dropout=0.0,
                 n_head=8,
                 Time_Input=False,
                 act='gelu',
                 mlp_ratio=1,
                 fun_dim=1,
                 out_dim=1,
                 slice_num=32,
                 ref=8,
                 unified_pos=False,
                 H=85,
                 W=85,
                 ):
        super(Model, self).__init__()
        self.__name__ = 'Transolver_2D'
        self.H = H
        self.W = W
        self.ref = ref
        self.unified_pos = unified_pos
        if self.unified_pos:
            # non-persistent buffer: follows .cuda()/.to() and stays out of the state_dict, like the
            # reference's plain attribute (which hard-codes .cuda(), :189,195)
            self.register_buffer("pos", self.get_grid(), persistent=False)
            self.preprocess = MLP(fun_dim + self.ref * self.ref, n_hidden * 2, n_hidden, n_layers=0, res=False, act=act)
        else:
            self.preprocess = MLP(fun_dim + space_dim, n_hidden * 2, n_hidden, n_layers=0, res=False, act=act)

        self.Time_Input = Time_Input
        self.n_hidden = n_hidden
        self.space_dim = space_dim
        if Time_Input:
            self.time_fc = nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.SiLU(), nn.Linear(n_hidden, n_hidden))

        self.blocks = nn.ModuleList([Transolver_block(num_heads=n_head, hidden_dim=n_hidden, dropout=dropout,
                                                      act=act, mlp_ratio=mlp_ratio, out_dim=out_dim,
                                                      slice_num=slice_num, H=H, W=W,
                                                      last_layer=(i == n_layers - 1))
                                     for i in range(n_layers)])
        self.initialize_weights()
        self.placeholder = nn.Parameter((1 / (n_hidden)) * torch.rand(n_hidden, dtype=torch.float))

    def initialize_weights(self):
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def get_grid(self, batchsize=1):
        """[batchsize, H, W, ref*ref]: Euclidean distance of every mesh point (linspace 0..1 per axis)
        to a ref x ref lattice; linspace in float64, arithmetic in float32 (as :183-200)."""
        H, W, ref = self.H, self.W, self.ref
        gy = torch.tensor(np.linspace(0, 1, H), dtype=torch.float)
        gx = torch.tensor(np.linspace(0, 1, W), dtype=torch.float)
        ry = torch.tensor(np.linspace(0, 1, ref), dtype=torch.float)
        rx = torch.tensor(np.linspace(0, 1, ref), dtype=torch.float)
        d0 = (gy[:, None] - ry[None, :]) ** 2            # H, ref
        d1 = (gx[:, None] - rx[None, :]) ** 2            # W, ref
        pos = torch.sqrt(d0[:, None, :, None] + d1[None, :, None, :]).reshape(1, H, W, ref * ref)
        return pos.repeat(batchsize, 1, 1, 1).contiguous()

    def forward(self, x, fx, T=None):
        B = x.shape[0]
        if self.unified_pos:
            x = self.pos.expand(B, -1, -1, -1).reshape(B, self.H * self.W, self.ref * self.ref)
        if fx is not None:
            fx = self.preprocess(torch.cat((x, fx), -1))
        else:
            fx = self.preprocess(x)
            fx = fx + self.placeholder[None, None, :]

        if T is not None:
            # [B,1] -> [B,1,C]; the reference repeats it over N before time_fc, which is the same
            # per-point linear map, so it is applied once and broadcast
            emb = timestep_embedding(T, self.n_hidden)
            emb = Fn.mlp(emb, None, 'silu', self.time_fc[0].weight, self.time_fc[0].bias,
                         self.time_fc[2].weight, self.time_fc[2].bias)
            fx = fx + emb

        for block in self.blocks:
            fx = block(fx)
        return fx
