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
