"""Transolver for point clouds / irregular meshes — drop-in for the reference's
model/Transolver_Irregular_Mesh.py `Model` (:74-158) and `Physics_Attention_Irregular_Mesh`
(model/Physics_Attention.py:6-57), the family exp_elas.py trains.  Same libpa2d kernels as the
structured path; the differences are exactly the reference's: Linear instead of 3x3-conv projections,
NO temperature clamp, `placeholder` always added, `unified_pos` computed from the input coordinates."""
import numpy as np
import torch
import torch.nn as nn

from .. import functional as Fn
from ._core import ACTIVATION, MLP, BlockBase, TransolverBase  # noqa: F401


class Physics_Attention_Irregular_Mesh(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., slice_num=64):
        super().__init__()
        if dim_head * heads != dim:
            raise NotImplementedError("HIP path needs heads*dim_head == dim (true for every reference model)")
        self.dim_head, self.heads, self.scale = dim_head, heads, dim_head ** -0.5
        self.softmax, self.dropout = nn.Softmax(dim=-1), nn.Dropout(dropout)
        self.temperature = nn.Parameter(torch.full([1, heads, 1, 1], 0.5))
        self.in_project_x = nn.Linear(dim, dim)
        self.in_project_fx = nn.Linear(dim, dim)
        self.in_project_slice = nn.Linear(dim_head, slice_num)
        nn.init.orthogonal_(self.in_project_slice.weight)
        self.to_q, self.to_k, self.to_v = (nn.Linear(dim_head, dim_head, bias=False) for _ in range(3))
        self.to_out = nn.Sequential(nn.Linear(dim, dim), nn.Dropout(dropout))
        self.engine = None

    def attention_parameters(self):
        return (self.temperature, self.in_project_x.weight, self.in_project_x.bias, self.in_project_fx.weight,
                self.in_project_fx.bias, self.in_project_slice.weight, self.in_project_slice.bias,
                self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight, self.to_out[0].bias)

    def forward(self, x, residual=None):
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path; refusing to ignore it")
        return Fn.physics_attention(x, residual, None, None, self.heads, self.attention_parameters(),
                                    engine=self.engine)          # H = W = None -> irregular


class Transolver_block(BlockBase):
    def __init__(self, num_heads, hidden_dim, dropout, act='gelu', mlp_ratio=4, last_layer=False, out_dim=1,
                 slice_num=32):
        super().__init__()
        attn = Physics_Attention_Irregular_Mesh(hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads,
                                                dropout=dropout, slice_num=slice_num)
        self._assemble(attn, hidden_dim, act, mlp_ratio, last_layer, out_dim)


class Model(TransolverBase):
    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False, act='gelu',
                 mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False):
        super().__init__()
        self.__name__ = 'Transolver_1D'
        self.ref, self.unified_pos, self.space_dim = ref, unified_pos, space_dim

        def make_block(is_last):
            return Transolver_block(num_heads=n_head, hidden_dim=n_hidden, dropout=dropout, act=act,
                                    mlp_ratio=mlp_ratio, last_layer=is_last, out_dim=out_dim, slice_num=slice_num)

        self._assemble(make_block, fun_dim + (ref * ref if unified_pos else space_dim), n_layers, n_hidden,
                       Time_Input, act)

    def get_grid(self, x, batchsize=1):
        """[B, N, ref*ref]: distance of every input point to a ref x ref lattice on the unit square."""
        lat = torch.tensor(np.linspace(0, 1, self.ref), dtype=torch.float, device=x.device)
        nodes = torch.stack(torch.meshgrid(lat, lat, indexing="ij"), dim=-1).reshape(1, 1, self.ref ** 2, 2)
        return (x[:, :, None, :] - nodes).square().sum(-1).sqrt().contiguous()

    def forward(self, x, fx, T=None):
        if self.unified_pos:
            x = self.get_grid(x, x.shape[0])
        z = self._embed(x, fx, always_placeholder=True)
        if T is not None:
            z = self._add_time(z, T)
        return self._run_blocks(z)
