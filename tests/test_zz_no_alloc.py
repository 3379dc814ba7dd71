"""GPU: the allocation-free decode entry point (nv_attn_decode_paged_static, split-key partial states and
arrival counters in a device array built into the library) against nv_attn_decode_paged with a caller
workspace: the same kernels on the same plan, so the outputs must be bit-identical, with the partial
states of split sequences combined by a second launch (the default) and inside the launch (where the
library's counters must be zero before the first call and after every call).  Two calls in a row per
form, the second on a smaller input, so that stale partial states or counters of the first call would
show."""
import pytest
import torch

from conftest import make_paged_problem
from oracle import ref_attention

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _both(dll, p, H_q, H_kv, bs):
    stream = torch.cuda.current_stream().cuda_stream
    B = p["cu_k"].numel() - 1
    nb = p["k_cache"].shape[0]
    common = (p["q"].data_ptr(), p["k_cache"].data_ptr(), p["v_cache"].data_ptr())
    meta = (p["cu_q"].data_ptr(), p["cu_k"].data_ptr(), p["block_tables"].data_ptr(), B, p["block_tables"].shape[1],
            p["max_k"], H_q, H_kv, 128, nb, bs, p["scale"], p["q"].stride(0))
    need = dll.nv_attn_decode_workspace_bytes(B, H_q, H_kv, 128, p["max_k"])
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device=DEV)
    o_ws = torch.full_like(p["q"], float("nan"))
    o_st = torch.full_like(p["q"], float("nan"))
    rc = dll.nv_attn_decode_paged(*common, o_ws.data_ptr(), *meta, o_ws.stride(0), ws.data_ptr(), ws.numel(), stream)
    assert rc == 0, dll.nv_last_error()
    # warm-up call: the per-device address lookup (and HIP's lazy loading of the code object) happen here
    rc = dll.nv_attn_decode_paged_static(*common, o_st.data_ptr(), *meta, o_st.stride(0), stream)
    assert rc == 0, dll.nv_last_error()
    torch.cuda.synchronize()
    o_st.fill_(float("nan"))
    torch.cuda.synchronize()
    # free device memory as the driver sees it (torch's own counters cannot see a hipMalloc inside the library)
    free_before = torch.cuda.mem_get_info()[0]
    torch_before = torch.cuda.memory_allocated()
    rc = dll.nv_attn_decode_paged_static(*common, o_st.data_ptr(), *meta, o_st.stride(0), stream)
    assert rc == 0, dll.nv_last_error()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free_before, "the allocation-free entry point changed the free device memory"
    assert torch.cuda.memory_allocated() == torch_before
    return o_ws, o_st, need


@pytest.mark.parametrize("single", [0, 1, -1])
def test_static_scratch_entry_point_matches_workspace_entry_point(single):
    assert torch.cuda.is_available()
    from nanovllm_amd import _lib
    dll = _lib.lib()
    cap = dll.nv_attn_decode_static_scratch_bytes()
    assert cap == 32 << 20
    H_q, H_kv, bs = 32, 8, 256
    _lib.set_tuning(decode_single_launch=single)
    try:
        # first call: long contexts, many key chunks per (sequence, kv head); second: fewer, shorter sequences
        for seed, n_ks in ((1, [5000, 8192, 300, 4097]), (2, [2049, 700])):
            nb = sum((n + bs - 1) // bs for n in n_ks) + 1
            p = make_paged_problem(seed, [1] * len(n_ks), n_ks, H_q, H_kv, bs, nb, device=DEV)
            o_ws, o_st, need = _both(dll, p, H_q, H_kv, bs)
            assert 16 < need <= cap                                   # the split path (scratch in use)
            assert torch.equal(o_ws.view(torch.int16), o_st.view(torch.int16))
            ref = ref_attention.paged_attention_ref(p["q"].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(),
                                                    p["cu_q"].cpu(), p["cu_k"].cpu(), p["block_tables"].cpu(), p["scale"])
            assert (o_st.float().cpu() - ref).abs().max().item() <= 2e-2
    finally:
        _lib.set_tuning()
    # an unsplit launch needs no scratch and still runs
    p = make_paged_problem(3, [1] * 300, [70] * 300, 16, 8, bs, 301, device=DEV)
    o_ws, o_st, need = _both(dll, p, 16, 8, bs)
    assert need == 16 and torch.equal(o_ws.view(torch.int16), o_st.view(torch.int16))


def test_static_scratch_entry_point_combines_small_launches_in_the_launch():
    """With at most 8 (sequence, kv head) pairs the allocation-free entry point combines the key chunks inside the
    launch by default (its counters live in the library's scratch: zero at load, put back to zero by every launch),
    while the workspace entry point runs its merge launch: the two combines sum in different orders, so the outputs
    agree to a bf16 ulp, and both meet the oracle under the scale-aware row bound.  Three calls in a row (long, short,
    long) would expose counters that were not reset."""
    from conftest import assert_rows_close
    from nanovllm_amd import _lib
    dll = _lib.lib()
    H_q, H_kv, bs = 32, 8, 256
    for seed, n_ks in ((11, [16384]), (12, [2049]), (13, [9000])):
        nb = sum((n + bs - 1) // bs for n in n_ks) + 1
        p = make_paged_problem(seed, [1], n_ks, H_q, H_kv, bs, nb, device=DEV)
        o_ws, o_st, need = _both(dll, p, H_q, H_kv, bs)
        assert need > 16                                              # split: partial states and a combine
        ref = ref_attention.paged_attention_ref(p["q"].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(),
                                                p["cu_q"].cpu(), p["cu_k"].cpu(), p["block_tables"].cpu(), p["scale"])
        assert_rows_close(o_st, ref, f"static, in-launch combine, n_k {n_ks}")
        assert_rows_close(o_ws, ref, f"workspace, merge launch, n_k {n_ks}")
        diff = (o_st.float() - o_ws.float()).abs()
        assert (diff <= o_ws.float().abs() * 2 ** -7 + 1e-30).all()


def test_static_scratch_entry_point_refuses_oversized_launches():
    from nanovllm_amd import _lib
    dll = _lib.lib()
    _lib.set_tuning(decode_chunk=64, decode_work_item=0)          # 64-key chunks: a huge partial-state footprint
    try:
        B, H_q, H_kv, n_k = 64, 32, 8, 16384
        assert dll.nv_attn_decode_workspace_bytes(B, H_q, H_kv, 128, n_k) > dll.nv_attn_decode_static_scratch_bytes()
        x = torch.zeros(B, H_q, 128, dtype=torch.bfloat16, device=DEV)
        kc = torch.zeros(1, 256, H_kv, 128, dtype=torch.bfloat16, device=DEV)
        cu = torch.arange(B + 1, dtype=torch.int32, device=DEV)
        bt = torch.zeros(B, 64, dtype=torch.int32, device=DEV)
        rc = dll.nv_attn_decode_paged_static(x.data_ptr(), kc.data_ptr(), kc.data_ptr(), x.data_ptr(), cu.data_ptr(),
                                             cu.data_ptr(), bt.data_ptr(), B, 64, n_k, H_q, H_kv, 128, 1, 256, 0.1,
                                             x.stride(0), x.stride(0), torch.cuda.current_stream().cuda_stream)
        assert rc == -3 and b"built-in scratch" in dll.nv_last_error()
    finally:
        _lib.set_tuning()
