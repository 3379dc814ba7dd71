"""Rotary position embedding (NeoX rotate-half), backed by a gfx950 HIP kernel.

Drop-in for the reference's nanovllm/layers/rope.py: ``RotaryPositionEmbedding(head_size,
rotary_dim, max_seq_len, base, device=None)`` owns the non-persistent fp32 buffer
``cos_sin_cache`` of shape ``[max_seq_len, 1, rotary_dim]`` (cos | sin) built exactly as at
rope.py:31-43; ``forward(positions, query, key)`` returns new rotated tensors (rope.py:45-56);
``get_rope`` is the cached factory (rope.py:59-69) and still refuses ``rope_scaling``.
"""
from __future__ import annotations

from functools import lru_cache

import torch
import torch.nn as nn

from .. import ops


def build_cos_sin_cache(rotary_dim: int, max_seq_len: int, base: float,
                        device: torch.device | None = None) -> torch.Tensor:
    """fp32 table [max_seq_len, 1, rotary_dim]: cos(pos * inv_freq) | sin(pos * inv_freq)."""
    exponent = torch.arange(0, rotary_dim, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / rotary_dim
    inv_freq = 1.0 / (base ** exponent)
    pos = torch.arange(max_seq_len, device=device, dtype=torch.float)
    angle = torch.einsum("i,j -> ij", pos, inv_freq)
    return torch.cat((angle.cos(), angle.sin()), dim=-1).unsqueeze_(1)


def apply_rotary_emb(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Host-visible helper with the reference's signature (rope.py:5-13).  The layer itself does
    not call it -- the rotation runs inside the HIP kernel -- it exists for callers that import it."""
    lo, hi = torch.chunk(x.float(), 2, dim=-1)
    return torch.cat((lo * cos - hi * sin, lo * sin + hi * cos), dim=-1).to(x.dtype)


class RotaryPositionEmbedding(nn.Module):
    def __init__(self, head_size: int, rotary_dim: int, max_seq_len: int, base: float,
                 device: torch.device | None = None):
        super().__init__()
        assert rotary_dim == head_size
        assert rotary_dim % 2 == 0, "dimension should be even in RotaryPositionEmbedding"
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self.register_buffer("cos_sin_cache", build_cos_sin_cache(rotary_dim, max_seq_len, base, device),
                             persistent=False)

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        return ops.rope(positions, query, key, self.cos_sin_cache)


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             rope_scaling: dict | None = None) -> RotaryPositionEmbedding:
    assert rope_scaling is None
    return RotaryPositionEmbedding(head_size, rotary_dim, max_position, base)
