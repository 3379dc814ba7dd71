"""Oracle (test infrastructure): RMSNorm, rotary embedding and SiluAndMul in plain torch on the CPU.

Follows nanovllm/layers/layernorm.py:79-100, nanovllm/layers/rope.py:5-56 and
nanovllm/layers/activation.py:10-16 of the reference.
"""
from __future__ import annotations

import torch


def rmsnorm_ref(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: torch.Tensor | None = None):
    """RMSNorm.forward (layernorm.py:79-100) as its compiled form evaluates it: fp32 throughout,
    one rounding to the input dtype after the weight multiply; with a residual the sum is
    formed in fp32 (:89), emitted rounded as the new residual (:90) and normalised unrounded."""
    dt = x.dtype
    xf = x.float()
    if residual is not None:
        xf = xf + residual.float()
        new_res = xf.to(dt)
    r = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)   # :95
    y = (xf * r * weight.float()).to(dt)                           # :96-98, single rounding
    return (y, new_res) if residual is not None else y


def silu_and_mul_ref(x: torch.Tensor) -> torch.Tensor:
    """SiluAndMul.forward (activation.py:10-16) as its compiled form evaluates it: the halves of
    the last dimension in fp32, silu(gate) * up, one rounding to the input dtype."""
    d = x.shape[-1] // 2
    g, u = x[..., :d].float(), x[..., d:].float()
    return (g * torch.sigmoid(g) * u).to(x.dtype)


def rmsnorm_ref_eager(x: torch.Tensor, weight: torch.Tensor, eps: float, residual: torch.Tensor | None = None):
    """The same forward as eager PyTorch evaluates the source text: the normalised value is
    rounded to the input dtype BEFORE the weight multiply (layernorm.py:98).  Differs from the
    compiled form by at most one ulp of the output dtype."""
    dt = x.dtype
    xf = x.float()
    if residual is not None:
        xf = xf + residual.float()
        new_res = xf.to(dt)
    r = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    y = (xf * r).to(dt) * weight
    return (y, new_res) if residual is not None else y


def llama3_inv_freq_ref(inv_freq: torch.Tensor, rope_scaling: dict) -> torch.Tensor:
    """The published Llama-3.1 ("llama3" rope_type) frequency rule, one frequency at a time in fp32
    scalars.  The reference refuses rope scaling (nanovllm/layers/rope.py:67 asserts
    ``rope_scaling is None``), so there is nothing of its own to follow: this restates the rule of
    the Llama-3.1 model card / Hugging Face ``rope_type="llama3"`` initialisation.  With
    wavelength = 2*pi / f and L = original_max_position_embeddings:
      wavelength > L / low_freq_factor   ->  f / factor            (long waves: slowed down)
      wavelength < L / high_freq_factor  ->  f                     (short waves: unchanged)
      otherwise                          ->  (1 - s) * f / factor + s * f,
                                             s = (L / wavelength - low) / (high - low)
    """
    import numpy as np
    f32 = np.float32
    factor, low, high = (f32(rope_scaling[k]) for k in ("factor", "low_freq_factor", "high_freq_factor"))
    L = f32(rope_scaling["original_max_position_embeddings"])
    low_wavelen, high_wavelen = L / low, L / high
    out = []
    for f in inv_freq.numpy().astype(np.float32):
        wavelen = f32(2 * np.pi) / f
        if wavelen > low_wavelen:
            out.append(f / factor)
        elif wavelen < high_wavelen:
            out.append(f)
        else:
            s = (L / wavelen - low) / (high - low)
            out.append((f32(1) - s) * f / factor + s * f)
    return torch.from_numpy(np.array(out, dtype=np.float32))


def build_cos_sin_cache(rotary_dim: int, max_seq_len: int, base: float, rope_scaling: dict | None = None) -> torch.Tensor:
    """rope.py:31-43: fp32 [max_seq_len, 1, rotary_dim] = cos(pos x inv_freq) | sin(...).
    `rope_scaling` (not supported by the reference) rescales inv_freq by llama3_inv_freq_ref first."""
    inv_freq = 1.0 / (base ** (torch.arange(0, rotary_dim, 2, dtype=torch.int64).to(torch.float) / rotary_dim))
    if rope_scaling is not None:
        kind = rope_scaling.get("rope_type", rope_scaling.get("type"))
        assert kind == "llama3", f"oracle: rope_scaling type {kind!r} has no restatement"
        inv_freq = llama3_inv_freq_ref(inv_freq, rope_scaling)
    pos = torch.arange(max_seq_len, dtype=torch.float)
    freqs = torch.einsum("i,j -> ij", pos, inv_freq)
    return torch.cat((freqs.cos(), freqs.sin()), dim=-1).unsqueeze(1)


def rope_ref(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, cos_sin_cache: torch.Tensor):
    """rope.py:46-56 with apply_rotary_emb :5-13 (fp32 math, cast back to the input dtype)."""
    cs = cos_sin_cache[positions]                       # [T, 1, D]
    cos, sin = cs.chunk(2, dim=-1)

    def rot(x):
        x1, x2 = torch.chunk(x.float(), 2, dim=-1)
        return torch.cat((x1 * cos - x2 * sin, x1 * sin + x2 * cos), dim=-1).to(x.dtype)

    return rot(query), rot(key)


def qknorm_rope_ref(positions, q, k, q_w, k_w, cos_sin_cache, eps):
    """q_norm, k_norm, rope chained as nanovllm/model/qwen3.py:67-71 does (each module rounds
    its output to bf16)."""
    if q_w is not None:
        q = rmsnorm_ref(q, q_w, eps)
        k = rmsnorm_ref(k, k_w, eps)
    return rope_ref(positions, q, k, cos_sin_cache)
