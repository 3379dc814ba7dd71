"""SiluAndMul with the reference's interface (nanovllm/layers/activation.py:6-16), computed by the
HIP kernel ``nv_silu_and_mul``: one launch instead of the silu and multiply kernels."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops


class SiluAndMul(nn.Module):
    def __init__(self):
        super().__init__()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.silu_and_mul(x)
