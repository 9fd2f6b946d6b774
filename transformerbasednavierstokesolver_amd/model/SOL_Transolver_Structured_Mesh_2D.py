"""Autoregressive ("unrolled") wrapper with the interface of the reference's
model/SOL_Transolver_Structured_Mesh_2D.py: construct it with the Transolver keywords plus `step` (how
many input channels one prediction replaces) and `look_ahead`; `forward(x, fx)` chains `self.n` model
calls, feeding each prediction back into the input window, and returns the last one.  Autograd runs
through the whole chain.  Attributes used by the reference's drivers: `.transolver_model`, `.n`, `.step`."""
import torch
import torch.nn as nn

from . import Transolver_Structured_Mesh_2D as _structured


class SOL_Transolver_Structured_Mesh_2D(nn.Module):
    def __init__(self, *model_args, step=1, look_ahead=5, **model_kwargs):
        super().__init__()
        self.transolver_model = _structured.Model(*model_args, **model_kwargs)
        self.n, self.step = look_ahead, step

    def forward(self, x, fx):
        prediction = None
        for _ in range(self.n):
            prediction = self.transolver_model(x, fx=fx)
            fx = torch.cat((fx[..., self.step:], prediction), dim=-1)     # drop the oldest, append the newest
        return prediction
