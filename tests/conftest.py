import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bf16(arr: np.ndarray) -> torch.Tensor:
    """int16 bit patterns -> bfloat16 tensor."""
    return torch.from_numpy(np.ascontiguousarray(arr)).view(torch.bfloat16)


def bits(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous().view(torch.int16)


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return load


def make_paged_problem(seed, n_qs, n_ks, H_q, H_kv, block_size, num_blocks, D=128, device="cpu", pad_cols=1):
    """Random paged attention problem with shuffled physical blocks and -1 padded tables."""
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(num_blocks, generator=g).tolist()
    need = [(n + block_size - 1) // block_size for n in n_ks]
    assert sum(need) <= num_blocks, "not enough blocks"
    width = max(max(need), 1) + pad_cols
    tables, nxt = [], 0
    for nb in need:
        tables.append(perm[nxt:nxt + nb] + [-1] * (width - nb))
        nxt += nb
    T = sum(n_qs)
    q = torch.randn(T, H_q, D, generator=g).bfloat16()
    k_cache = torch.randn(num_blocks, block_size, H_kv, D, generator=g).bfloat16()
    v_cache = torch.randn(num_blocks, block_size, H_kv, D, generator=g).bfloat16()
    cu_q = torch.tensor([0] + np.cumsum(n_qs).tolist(), dtype=torch.int32)
    cu_k = torch.tensor([0] + np.cumsum(n_ks).tolist(), dtype=torch.int32)
    p = dict(q=q, k_cache=k_cache, v_cache=v_cache, cu_q=cu_q, cu_k=cu_k,
             block_tables=torch.tensor(tables, dtype=torch.int32), scale=D ** -0.5,
             max_q=max(n_qs) if n_qs else 0, max_k=max(n_ks) if n_ks else 0)
    if device != "cpu":
        p = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in p.items()}
    return p


# ------------------------------------------------------------------ scale-aware attention bound
C_ROW = 0.03        # a correct kernel stays below about a third of this (bf16 output rounding of a 3-sigma
                    # element is 0.006 rms; bf16 probabilities add about 0.002 rms)


def bf16_ulp(x: float) -> float:
    """Spacing of bfloat16 numbers at magnitude x (8 significant bits)."""
    import math
    return 2.0 ** (math.floor(math.log2(max(abs(x), 2.0 ** -126))) - 7)


def row_bounds(ref: torch.Tensor, c: float = C_ROW) -> torch.Tensor:
    """Per output row (last dimension = head_dim): max(one bf16 ulp of max|ref|, c * rms(ref)).
    A row that averages n random value rows has rms ~ sqrt(e / n), so a fixed absolute bound says less
    and less as contexts grow; this one scales with the answer."""
    ref = ref.float()
    amax = ref.abs().amax(dim=-1)
    ulp = torch.exp2(torch.floor(torch.log2(amax.clamp_min(2.0 ** -126))) - 7)
    rms = ref.pow(2).mean(dim=-1).sqrt()
    return torch.maximum(ulp, c * rms)


def assert_rows_close(out: torch.Tensor, ref: torch.Tensor, name, c: float = C_ROW, abs_tol: float = 2e-2) -> float:
    """Both bars: the fixed max-abs tolerance and the per-row scale-aware one.  Returns the worst
    ratio err / bound (a correct kernel is expected below about 1/3)."""
    out, ref = out.float().cpu(), ref.float().cpu()
    assert out.shape == ref.shape, (name, out.shape, ref.shape)
    err = (out - ref).abs().amax(dim=-1)
    assert torch.isfinite(out).all(), (name, "non-finite output")
    assert err.max().item() <= abs_tol, (name, "max-abs", err.max().item())
    ratio = err / row_bounds(ref, c)
    worst = ratio.max().item()
    assert worst <= 1.0, (name, "row error / scale-aware bound", worst, "at", int(ratio.argmax()))
    return worst


def assert_bound_is_sensitive(ref: torch.Tensor, ref_damaged: torch.Tensor, name, c: float = C_ROW) -> float:
    """The check must be able to fail: the oracle's own answer with one 256-key block taken out of the
    sequence (``ref_damaged``) has to violate the bound on every row given.  Returns the smallest ratio."""
    ref, ref_damaged = ref.float().cpu(), ref_damaged.float().cpu()
    err = (ref_damaged - ref).abs().amax(dim=-1)
    ratio = err / row_bounds(ref, c)
    least = ratio.min().item()
    assert least > 1.0, (name, "insensitive: dropping a key block stays inside the bound", least)
    return least
