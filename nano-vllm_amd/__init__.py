"""MI355X-native attention hot path with the module contract of pradsgit/nano-vllm.

Layers (same constructor / forward signatures as the reference's ``nanovllm.layers``):
``Attention``, ``RMSNorm``, ``RotaryPositionEmbedding`` / ``get_rope``, ``SiluAndMul``; per-step metadata
travels in ``CacheContext`` via ``set_context`` / ``get_context``.  All device work runs
in hand-written gfx950 HIP kernels reached through the C ABI in ``include/nanovllm_hip.h``;
there is no CPU or eager fallback: a missing library is an error.
"""
from .utils.cache_context import CacheContext, set_context, get_context  # noqa: F401
from .layers.attention import Attention  # noqa: F401
from .layers.layernorm import RMSNorm  # noqa: F401
from .layers.rope import RotaryPositionEmbedding, get_rope, apply_rotary_emb  # noqa: F401
from .layers.activation import SiluAndMul  # noqa: F401

__all__ = [
    "Attention", "RMSNorm", "RotaryPositionEmbedding", "get_rope", "apply_rotary_emb", "SiluAndMul",
    "CacheContext", "set_context", "get_context",
]
