"""Shared building blocks of the MI355X Transolver models.

The reference keeps one copy of MLP / block / Model per mesh family (model/Transolver_*.py); here the
families differ only in the attention module handed to `TransolverBase._assemble`, so everything else
lives once in this file.  Attribute names (`preprocess`, `time_fc`, `blocks.i.{ln_1,Attn,ln_2,mlp,ln_3,
mlp2}`, `placeholder`) are the reference's, because they ARE the state_dict contract."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as Fn
from .Embedding import timestep_embedding

# activation names accepted by the reference's ACTIVATION table (model/Transolver_Structured_Mesh_2D.py:9-10);
# its 'leaky_relu' entry is an nn.Module instance instead of a class and cannot be constructed there either
ACTIVATION = {'gelu': nn.GELU, 'tanh': nn.Tanh, 'sigmoid': nn.Sigmoid, 'relu': nn.ReLU,
              'softplus': nn.Softplus, 'ELU': nn.ELU, 'silu': nn.SiLU}
HEAD_KERNEL_MAX_OUT = 8        # widest output the dedicated head kernel handles


def pad_contraction(x, w):
    """libpa2d GEMMs load 16 bytes at a time: zero-pad the contraction length to a multiple of 4."""
    rem = w.shape[1] % 4
    if rem:
        x, w = F.pad(x, (0, 4 - rem)), F.pad(w, (0, 4 - rem))
    return x, w


class MLP(nn.Module):
    """Linear -> act -> [hidden Linear -> act (+ skip)] x n_layers -> Linear (reference MLP, :13-38).
    The activation modules inside the Sequentials are parameter-free markers; the arithmetic runs in
    the GEMM epilogues of libpa2d."""

    def __init__(self, n_input, n_hidden, n_output, n_layers=1, act='gelu', res=True):
        super().__init__()
        if act not in ACTIVATION:
            raise NotImplementedError
        self.act_name, self.res = act, res
        self.engine = None              # GEMM engine of this module's kernels (None = pa2d_default_engine())
        self.n_input, self.n_hidden, self.n_output, self.n_layers = n_input, n_hidden, n_output, n_layers
        make = ACTIVATION[act]
        self.linear_pre = nn.Sequential(nn.Linear(n_input, n_hidden), make())
        self.linear_post = nn.Linear(n_hidden, n_output)
        self.linears = nn.ModuleList(nn.Sequential(nn.Linear(n_hidden, n_hidden), make()) for _ in range(n_layers))

    def forward(self, x, residual=None):
        first, last = self.linear_pre[0], self.linear_post
        x, w_first = pad_contraction(x, first.weight)
        if not self.linears:            # the only shape the Transolver blocks use: one fused pair of GEMMs
            return Fn.mlp(x, residual, self.act_name, w_first, first.bias, last.weight, last.bias, engine=self.engine)
        h = Fn.linear(x, w_first, first.bias, self.act_name, engine=self.engine)
        for hidden in self.linears:
            y = Fn.linear(h, hidden[0].weight, hidden[0].bias, self.act_name, engine=self.engine)
            h = y + h if self.res else y
        out = Fn.linear(h, last.weight, last.bias, None, engine=self.engine)
        return out if residual is None else out + residual


class BlockBase(nn.Module):
    """pre-LN residual block: fx += Attn(LN1 fx); fx += MLP(LN2 fx); last block: head(LN3 fx)."""

    def _assemble(self, attn, hidden_dim, act, mlp_ratio, last_layer, out_dim):
        self.last_layer = last_layer
        self.engine = None
        self.ln_1 = nn.LayerNorm(hidden_dim)
        self.Attn = attn
        self.ln_2 = nn.LayerNorm(hidden_dim)
        self.mlp = MLP(hidden_dim, hidden_dim * mlp_ratio, hidden_dim, n_layers=0, res=False, act=act)
        if last_layer:
            self.ln_3 = nn.LayerNorm(hidden_dim)
            self.mlp2 = nn.Linear(hidden_dim, out_dim)

    def forward(self, fx):
        attn, mlp = self.Attn, self.mlp
        if attn.training and attn.dropout.p > 0:
            raise NotImplementedError("dropout > 0 is not implemented in the HIP path; refusing to ignore it")
        # each residual branch (LayerNorm -> sub-layer -> + fx) is one autograd node
        fx = Fn.attn_branch(fx, self.ln_1.weight, self.ln_1.bias, getattr(attn, "H", None), getattr(attn, "W", None),
                            attn.heads, attn.attention_parameters(), engine=self.engine)
        pre, post = mlp.linear_pre[0], mlp.linear_post
        if mlp.linears or pre.weight.shape[1] % 4:      # generic MLP shapes keep the unfused route
            fx = mlp(Fn.layer_norm(fx, self.ln_2.weight, self.ln_2.bias), residual=fx)
        else:
            fx = Fn.mlp_branch(fx, self.ln_2.weight, self.ln_2.bias, mlp.act_name, pre.weight, pre.bias,
                               post.weight, post.bias, engine=self.engine)
        if not self.last_layer:
            return fx
        z = Fn.layer_norm(fx, self.ln_3.weight, self.ln_3.bias)
        if self.mlp2.out_features <= HEAD_KERNEL_MAX_OUT:
            return Fn.head(z, self.mlp2.weight, self.mlp2.bias)          # fp32 out, whatever the storage type
        return Fn.linear(z, self.mlp2.weight, self.mlp2.bias, None, engine=self.engine).float()


class TransolverBase(nn.Module):
    def _assemble(self, make_block, in_features, n_layers, n_hidden, Time_Input, act):
        self.Time_Input, self.n_hidden = Time_Input, n_hidden
        self.engine = None
        self.preprocess = MLP(in_features, n_hidden * 2, n_hidden, n_layers=0, res=False, act=act)
        if Time_Input:
            self.time_fc = nn.Sequential(nn.Linear(n_hidden, n_hidden), nn.SiLU(), nn.Linear(n_hidden, n_hidden))
        self.blocks = nn.ModuleList(make_block(i == n_layers - 1) for i in range(n_layers))
        self.initialize_weights()
        # created after the init pass, uniform in [0, 1/C) like the reference (…_2D.py:167)
        self.placeholder = nn.Parameter(torch.rand(n_hidden, dtype=torch.float) / n_hidden)

    def initialize_weights(self):
        self.apply(self._init_weights)

    def _init_weights(self, m):
        """Linear: trunc_normal(std .02) / zero bias; LayerNorm: 1 / 0; Conv2d keeps PyTorch's default."""
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            nn.init.ones_(m.weight)
            nn.init.zeros_(m.bias)

    def _embed(self, x, fx, always_placeholder):
        if fx is not None:
            z = self.preprocess(torch.cat((x, fx), -1))
            return z + self.placeholder[None, None, :] if always_placeholder else z
        return self.preprocess(x) + self.placeholder[None, None, :]

    def _add_time(self, z, T):
        """T: [B,1].  The reference repeats the embedding over the N points before `time_fc`
        (…_2D.py:212-215); that is the same per-point linear map, so it is applied once and broadcast."""
        fc = self.time_fc
        emb = timestep_embedding(T, self.n_hidden)
        return z + Fn.mlp(emb, None, 'silu', fc[0].weight, fc[0].bias, fc[2].weight, fc[2].bias, engine=self.engine)

    def set_engine(self, engine):
        """Select the GEMM engine ("f32" | "split" | "bf16" | 0 | 1 | 2 | None = environment default) for every
        kernel of THIS model; other models in the process keep theirs (the library holds no engine state)."""
        from .. import ops
        eng = None if engine is None else ops.resolve_engine(engine)
        for m in self.modules():
            if hasattr(m, "engine"):
                m.engine = eng
        return self

    def _run_blocks(self, z):
        from .. import ops
        if self.engine == ops.ENGINE_BF16S:
            # bf16 storage: from here on every activation, saved tensor and inter-kernel gradient is bf16 (the cast is
            # the one fp32 -> bf16 boundary; its autograd node casts the gradient back for the fp32 input embedding)
            z = z.to(torch.bfloat16)
        for block in self.blocks:
            z = block(z)
        return z
