"""Physics-Attention for structured 2-D meshes, MI355X-native.

Same constructor, parameter names/shapes and `forward(x[B,N,C]) -> [B,N,C]` as the reference class
of the same name (model/Physics_Attention.py:60-119), but the forward/backward run as a handful of
hand-written HIP kernels (libpa2d): one implicit-GEMM for both 3x3 projections on the NHWC tensor
(no permute copies), MFMA slice-softmax-scatter, an in-LDS token attention, a de-slice that
recomputes the slice weights, and an MFMA GEMM for to_out with bias/residual epilogue.
The nn.Conv2d / nn.Linear sub-modules are parameter containers only (identical state_dict keys and
default initialisation); their own forward is never called.
"""
import torch
import torch.nn as nn

from .. import functional as Fn


class Physics_Attention_Structured_Mesh_2D(nn.Module):
    def __init__(self, dim, heads=8, dim_head=64, dropout=0., slice_num=64, H=101, W=31, kernel=3):
        super().__init__()
        inner_dim = dim_head * heads
        if kernel != 3:
            raise NotImplementedError("the HIP path implements the 3x3 projection used by every reference script")
        if inner_dim != dim:
            raise NotImplementedError("HIP path needs heads*dim_head == dim (true for every reference model)")
        self.dim_head = dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.softmax = nn.Softmax(dim=-1)
        self.dropout = nn.Dropout(dropout)
        self.temperature = nn.Parameter(torch.ones([1, heads, 1, 1]) * 0.5)
        self.H = H
        self.W = W
        self.engine = None      # GEMM engine (None = pa2d_default_engine()); TransolverBase.set_engine sets it

        self.in_project_x = nn.Conv2d(dim, inner_dim, kernel, 1, kernel // 2)
        self.in_project_fx = nn.Conv2d(dim, inner_dim, kernel, 1, kernel // 2)
        self.in_project_slice = nn.Linear(dim_head, slice_num)
        torch.nn.init.orthogonal_(self.in_project_slice.weight)
        self.to_q = nn.Linear(dim_head, dim_head, bias=False)
        self.to_k = nn.Linear(dim_head, dim_head, bias=False)
        self.to_v = nn.Linear(dim_head, dim_head, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def attention_parameters(self):
        return (self.temperature, self.in_project_x.weight, self.in_project_x.bias, self.in_project_fx.weight,
                self.in_project_fx.bias, self.in_project_slice.weight, self.in_project_slice.bias,
                self.to_q.weight, self.to_k.weight, self.to_v.weight, self.to_out[0].weight, self.to_out[0].bias)

    def forward(self, x, residual=None):
        """x: [B, N=H*W, C].  `residual` (extension): added in the to_out epilogue (block uses it)."""
        if self.training and self.dropout.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path (every reference script "
                                      "uses --dropout 0.0); refusing to silently ignore it")
        B, N, C = x.shape
        if N != self.H * self.W:
            raise RuntimeError(f"shape '[{B}, {self.H}, {self.W}, {C}]' is invalid for input of size {x.numel()}")
        return Fn.physics_attention(x, residual, self.H, self.W, self.heads, self.attention_parameters(),
                                    engine=self.engine)
