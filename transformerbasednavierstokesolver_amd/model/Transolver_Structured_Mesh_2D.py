"""Transolver for structured 2-D meshes — the drop-in for the reference module of the same name:
`Model` keeps its 15 keyword parameters and defaults (model/Transolver_Structured_Mesh_2D.py:123-139),
`forward(x, fx, T=None)` (:202-220) and the 169-tensor state_dict; the arithmetic runs on libpa2d.
Shared pieces (MLP, block, init, embedding) are in `_core.py`."""
import numpy as np
import torch

from ._core import ACTIVATION, MLP, BlockBase, TransolverBase  # noqa: F401  (MLP/ACTIVATION re-exported like the reference)
from .Physics_Attention import Physics_Attention_Structured_Mesh_2D


class Transolver_block(BlockBase):
    def __init__(self, num_heads, hidden_dim, dropout, act='gelu', mlp_ratio=4, last_layer=False, out_dim=1,
                 slice_num=32, H=85, W=85):
        super().__init__()
        attn = Physics_Attention_Structured_Mesh_2D(hidden_dim, heads=num_heads, dim_head=hidden_dim // num_heads,
                                                    dropout=dropout, slice_num=slice_num, H=H, W=W)
        self._assemble(attn, hidden_dim, act, mlp_ratio, last_layer, out_dim)


class Model(TransolverBase):
    def __init__(self, space_dim=1, n_layers=5, n_hidden=256, dropout=0.0, n_head=8, Time_Input=False, act='gelu',
                 mlp_ratio=1, fun_dim=1, out_dim=1, slice_num=32, ref=8, unified_pos=False, H=85, W=85):
        super().__init__()
        self.__name__ = 'Transolver_2D'
        self.H, self.W, self.ref, self.unified_pos, self.space_dim = H, W, ref, unified_pos, space_dim
        if unified_pos:
            # non-persistent buffer: follows .cuda()/.to() and stays out of the state_dict, like the
            # reference's plain attribute (whose get_grid hard-codes .cuda(), :189,195)
            self.register_buffer("pos", self.get_grid(), persistent=False)

        def make_block(is_last):
            return Transolver_block(num_heads=n_head, hidden_dim=n_hidden, dropout=dropout, act=act,
                                    mlp_ratio=mlp_ratio, last_layer=is_last, out_dim=out_dim, slice_num=slice_num,
                                    H=H, W=W)

        self._assemble(make_block, fun_dim + (ref * ref if unified_pos else space_dim), n_layers, n_hidden,
                       Time_Input, act)

    def get_grid(self, batchsize=1):
        """[batchsize, H, W, ref*ref]: Euclidean distance of mesh point (i/(H-1), j/(W-1)) to the lattice
        point (k/(ref-1), l/(ref-1)), feature index k*ref+l; linspace in float64, arithmetic in float32."""
        axis = lambda n: torch.tensor(np.linspace(0, 1, n), dtype=torch.float)
        rows, cols, lat = axis(self.H), axis(self.W), axis(self.ref)
        dr2 = (rows[:, None] - lat[None, :]) ** 2           # H, ref
        dc2 = (cols[:, None] - lat[None, :]) ** 2           # W, ref
        pos = torch.sqrt(dr2[:, None, :, None] + dc2[None, :, None, :])
        return pos.reshape(1, self.H, self.W, self.ref ** 2).repeat(batchsize, 1, 1, 1).contiguous()

    def forward(self, x, fx, T=None):
        if self.unified_pos:      # the coordinates in `x` are ignored (only the batch size is used)
            x = self.pos.expand(x.shape[0], -1, -1, -1).reshape(x.shape[0], self.H * self.W, self.ref ** 2)
        z = self._embed(x, fx, always_placeholder=False)
        if T is not None:
            z = self._add_time(z, T)
        return self._run_blocks(z)
