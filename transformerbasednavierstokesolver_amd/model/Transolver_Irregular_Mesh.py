"""Transolver for point clouds / irregular meshes — drop-in for the reference's
model/Transolver_Irregular_Mesh.py (`Model` :74-158, `Transolver_block` :40-71) and
`Physics_Attention_Irregular_Mesh` (model/Physics_Attention.py:6-57), used by exp_elas.py.
Same libpa2d kernels as the structured path; the differences are exactly the reference's:
Linear instead of 3x3-conv projections, NO temperature clamp, `placeholder` always added, and
`unified_pos` computed from the input coordinates."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from .Embedding import timestep_embedding
from .Transolver_Structured_Mesh_2D import MLP, ACTIVATION  # noqa: F401  (same MLP class as the reference's copy)


class Physics_Attention_Irregular_Mesh(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., slice_num=64):
        super().__init__()
        inner_dim = dim_head * heads
        if inner_dim != dim:
            raise NotImplementedError("HIP path needs heads*dim_head == dim (true for every reference model)")
        self.dim_head = dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.temperature = nn.Parameter(torch.ones([1, heads, 1, 1]) * 0.5)
        self.in_project_x = nn.Linear(dim, inner_dim)
        self.in_project_fx = nn.Linear(dim, inner_dim)
        self.in_project_slice = nn.Linear(dim_head, slice_num)
        torch.nn.init.orthogonal_(self.in_project_slice.weight)
        self.to_q = nn.Linear(dim_head, dim_head, bias=False)
        self.to_k = nn.Linear(dim_head, dim_head, bias=False)
        self.to_v = nn.Linear(dim_head, dim_head, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def _params(self):
        return (self.temperature, self.in_project_x.weight, self.in_project_x.bias, self.in_project_fx.weight,
                self.in_project_fx.bias, self.in_project_slice.weight, self.in_project_slice.bias,
                self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight, self.to_out[0].bias)

    def forward(self, x, residual=None):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path; refusing to ignore it")
        return Fn.physics_attention(x, residual, None, None, self.heads, self._params())


class Transolver_block(nn.Module):
    def __init__(self, num_heads, hidden_dim, dropout, act='gelu', mlp_ratio=4, last_layer=False, out_dim=1,
                 slice_num=32):
        super().__init__()
        self.last_layer = last_layer
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.Attn = Physics_Attention_Irregular_Mesh(hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads,
                                                     dropout=dropout, slice_num=slice_num)
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
    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False, act='gelu',
                 mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False):
        super(Model, self).__init__()
        self.__name__ = 'Transolver_1D'
        self.ref = ref
        self.unified_pos = unified_pos
        self.Time_Input = Time_Input
        self.n_hidden = n_hidden
        self.space_dim = space_dim
        in_dim = fun_dim + (self.ref * self.ref if self.unified_pos else space_dim)
        self.preprocess = MLP(in_dim, n_hidden * 2, n_hidden, n_layers=0, res=False, act=act)
        if Time_Input:
            self.time_fc = nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.SiLU(), nn.Linear(n_hidden, n_hidden))
        self.blocks = nn.ModuleList([Transolver_block(num_heads=n_head, hidden_dim=n_hidden, dropout=dropout, act=act,
                                                      mlp_ratio=mlp_ratio, out_dim=out_dim, slice_num=slice_num,
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

    def get_grid(self, x, batchsize=1):
        """[B, N, ref*ref]: distance of every input point (first two coordinates) to a ref x ref lattice."""
        r = torch.tensor(np.linspace(0, 1, self.ref), dtype=torch.float, device=x.device)
        gx = r.reshape(self.ref, 1).expand(self.ref, self.ref)
        gy = r.reshape(1, self.ref).expand(self.ref, self.ref)
        grid_ref = torch.stack((gx, gy), dim=-1).reshape(1, self.ref * self.ref, 2)
        return torch.sqrt(torch.sum((x[:, :, None, :] - grid_ref[:, None, :, :]) ** 2, dim=-1)).contiguous()

    def forward(self, x, fx, T=None):
        if self.unified_pos:
            x = self.get_grid(x, x.shape[0])
        if fx is not None:
            fx = self.preprocess(torch.cat((x, fx), -1))
        else:
            fx = self.preprocess(x)
        fx = fx + self.placeholder[None, None, :]
        if T is not None:
            emb = timestep_embedding(T, self.n_hidden)
            emb = Fn.mlp(emb, None, 'silu', self.time_fc[0].weight, self.time_fc[0].bias,
                         self.time_fc[2].weight, self.time_fc[2].bias)
            fx = fx + emb
        for block in self.blocks:
            fx = block(fx)
        return fx
