"""RMSNorm with optional fused residual add, backed by gfx950 HIP kernels.

Drop-in for the reference's live ``RMSNorm`` (nanovllm/layers/layernorm.py:60-100):
``RMSNorm(hidden_size, eps=1e-6, dtype=None, device=None)`` with a ``weight`` parameter, and
``forward(x, residual=None)`` returning ``y`` or ``(y, new_residual)``.  Arithmetic is fp32
with ONE rounding to bf16 after the weight multiply (what the reference's compiled forward
produces); ``new_residual = bf16(f32(x) + f32(residual))``.  Inputs are never modified.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class RMSNorm(nn.Module):
    def __init__(self, hidden_size: int, eps: float = 1e-6, dtype: torch.dtype | None = None,
                 device: torch.device | None = None):
        super().__init__()
        self.hidden_size = hidden_size
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(hidden_size, dtype=dtype, device=device))

    def forward(self, x: torch.Tensor, residual: torch.Tensor | None = None):
        if x.shape[-1] != self.hidden_size:
            raise ValueError(f"last dim {x.shape[-1]} != hidden_size {self.hidden_size}")
        if residual is None:
            return ops.rmsnorm(x, self.weight, self.eps)
        return ops.add_rmsnorm(x, residual, self.weight, self.eps)
