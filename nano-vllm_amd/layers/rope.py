"""Rotary position embedding (NeoX rotate-half), backed by a gfx950 HIP kernel.

Drop-in for the reference's nanovllm/layers/rope.py: ``RotaryPositionEmbedding(head_size,
rotary_dim, max_seq_len, base, device=None)`` owns the non-persistent fp32 buffer
``cos_sin_cache`` of shape ``[max_seq_len, 1, rotary_dim]`` (cos | sin) built exactly as at
rope.py:31-43; ``forward(positions, query, key)`` returns new rotated tensors (rope.py:45-56);
``get_rope`` is the cached factory (rope.py:59-69) and still refuses ``rope_scaling``;
``make_rope`` additionally builds Llama-3.1 ("llama3") frequency-scaled tables.
"""
from __future__ import annotations

from functools import lru_cache

import torch
import torch.nn as nn

from .. import ops


def llama3_scale_inv_freq(inv_freq: torch.Tensor, factor: float, low_freq_factor: float,
                          high_freq_factor: float, original_max_position_embeddings: int) -> torch.Tensor:
    """Llama-3.1 frequency scaling (the published "llama3" rope_type rule): wavelengths longer than
    original/low_freq_factor are divided by `factor`, shorter than original/high_freq_factor are
    kept, and the band in between is interpolated smoothly.  The reference cannot run such models
    (its get_rope asserts rope_scaling is None, nanovllm/layers/rope.py:67)."""
    low_wavelen = original_max_position_embeddings / low_freq_factor
    high_wavelen = original_max_position_embeddings / high_freq_factor
    wavelen = 2 * torch.pi / inv_freq
    scaled = torch.where(wavelen > low_wavelen, inv_freq / factor, inv_freq)
    smooth = (original_max_position_embeddings / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
    mid = (1 - smooth) * inv_freq / factor + smooth * inv_freq
    band = (wavelen <= low_wavelen) & (wavelen >= high_wavelen)
    return torch.where(band, mid, scaled)


def build_cos_sin_cache(rotary_dim: int, max_seq_len: int, base: float,
                        device: torch.device | None = None, rope_scaling: dict | None = None) -> torch.Tensor:
    """fp32 table [max_seq_len, 1, rotary_dim]: cos(pos * inv_freq) | sin(pos * inv_freq)."""
    exponent = torch.arange(0, rotary_dim, 2, dtype=torch.int64).to(device=device, dtype=torch.float) / rotary_dim
    inv_freq = 1.0 / (base ** exponent)
    if rope_scaling is not None:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
        if kind != "llama3":
            raise NotImplementedError(f"rope_scaling type {kind!r} is not supported")
        inv_freq = llama3_scale_inv_freq(inv_freq, rope_scaling["factor"], rope_scaling["low_freq_factor"],
                                         rope_scaling["high_freq_factor"],
                                         rope_scaling["original_max_position_embeddings"])
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
                 device: torch.device | None = None, rope_scaling: dict | None = None):
        super().__init__()
        assert rotary_dim == head_size
        assert rotary_dim % 2 == 0, "dimension should be even in RotaryPositionEmbedding"
        self.head_size = head_size
        self.rotary_dim = rotary_dim
        self._table_args = (rotary_dim, max_seq_len, base, rope_scaling)
        self.register_buffer("cos_sin_cache",
                             build_cos_sin_cache(rotary_dim, max_seq_len, base, device, rope_scaling),
                             persistent=False)

    def _apply(self, fn, recurse=True):
        """The table stays fp32 whatever dtype the surrounding model is cast to (`model.to(bf16)`
        would otherwise round it): after a dtype-changing cast it is rebuilt on the new device."""
        super()._apply(fn, recurse)
        if self.cos_sin_cache.dtype != torch.float32:
            d, n, base, scaling = self._table_args
            self.cos_sin_cache = build_cos_sin_cache(d, n, base, None, scaling).to(self.cos_sin_cache.device)
        return self

    def forward(self, positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor):
        return ops.rope(positions, query, key, self.cos_sin_cache)


_scaled_ropes: dict = {}


def make_rope(head_size: int, max_position: int, base: float, rope_scaling: dict | None = None):
    """Shared rotary module per (shape, base, scaling).  Without scaling this is `get_rope`; with a
    llama3 scaling dict (which `get_rope` refuses, like the reference) a scaled table is built."""
    if rope_scaling is None:
        return get_rope(head_size, head_size, max_position, base)
    key = (head_size, max_position, base, tuple(sorted(rope_scaling.items())))
    if key not in _scaled_ropes:
        _scaled_ropes[key] = RotaryPositionEmbedding(head_size, head_size, max_position, base,
                                                     rope_scaling=rope_scaling)
    return _scaled_ropes[key]


@lru_cache(1)
def get_rope(head_size: int, rotary_dim: int, max_position: int, base: float,
             rope_scaling: dict | None = None) -> RotaryPositionEmbedding:
    assert rope_scaling is None
    return RotaryPositionEmbedding(head_size, rotary_dim, max_position, base)
