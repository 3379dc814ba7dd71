"""GPU: decode attention on skewed batches (a few long sequences among many short ones) with the host's choice
of keys per work item (nv_attn_decode_paged_chunked / nv_attn_decode_paged_fused_chunked,
nv_attn_decode_suggest_chunk): the long sequences are cut into many key ranges and merged, the short ones are
written by the attention launch directly.  Checked against the oracle under the scale-aware row bound, against
the unhinted launch, and through the engine (eager step and captured graph of the same step)."""
import math

import pytest
import torch

from conftest import assert_rows_close, make_paged_problem
from oracle import ref_attention
from test_gpu_fused_decode import _assert_same, _step

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def hip():
    from nanovllm_amd import _lib, ops
    _lib.set_tuning()
    yield ops
    _lib.set_tuning()


def _lengths(n_short, longs, seed):
    g = torch.Generator().manual_seed(seed)
    short = torch.randint(3, 700, (n_short,), generator=g).tolist()
    ctx = short + list(longs)
    order = torch.randperm(len(ctx), generator=g).tolist()       # the long ones anywhere in the batch
    return [ctx[i] for i in order]


@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (32, 8), (64, 8), (8, 1)])
@pytest.mark.parametrize("chunk", [256, 512, 1088])
def test_hinted_decode_against_the_oracle(hip, H_q, H_kv, chunk):
    ctx = _lengths(40, [5000, 2049, 1], seed=chunk + H_q)
    bs = 256
    nb = sum(math.ceil(c / bs) for c in ctx) + 2
    p = make_paged_problem(chunk, [1] * len(ctx), ctx, H_q, H_kv, bs, nb, device=DEV)
    ref = ref_attention.paged_attention_ref(p["q"].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(), p["cu_q"].cpu(),
                                            p["cu_k"].cpu(), p["block_tables"].cpu(), p["scale"])
    plain = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"], p["scale"])
    for wi in (-1, 0, 1):
        from nanovllm_amd import _lib
        _lib.set_tuning(decode_work_item=wi)
        try:
            o = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"],
                                      p["scale"], cu_seqlens_q=p["cu_q"], chunk_keys=chunk)
        finally:
            _lib.set_tuning()
        torch.cuda.synchronize()
        assert_rows_close(o, ref, f"H{H_q}/{H_kv} chunk {chunk} work item {wi}")
        # the same keys in another grouping: the unhinted launch's rows to fp32 summation order
        assert (o.float() - plain.float()).abs().max().item() <= 2e-2


@pytest.mark.parametrize("H_q,H_kv", [(16, 8), (32, 8), (64, 8)])
def test_hinted_one_launch_step_equals_the_two_launches(hip, H_q, H_kv):
    """The one-launch form with the same hint: every cache byte equal, outputs equal up to fp32 contraction."""
    ctx = _lengths(30, [4097, 3000], seed=H_q)
    p, q, k, v, q_w, k_w, cos_sin, pos, slots = _step(21, ctx, H_q, H_kv, 256, True)
    chunk = 512
    kc1, vc1 = p["k_cache"].clone(), p["v_cache"].clone()
    q_rot, _ = hip.qknorm_rope_store(pos, q, k, v, q_w, k_w, cos_sin, 1e-6, slots, kc1, vc1, want_k=False)
    o1 = hip.attn_decode_paged(q_rot, kc1, vc1, p["cu_k"], p["block_tables"], p["max_k"], p["scale"], chunk_keys=chunk)
    kc2, vc2 = p["k_cache"].clone(), p["v_cache"].clone()
    o2 = hip.attn_decode_paged_fused(pos, q, k, v, q_w, k_w, cos_sin, 1e-6, slots, kc2, vc2, p["cu_k"],
                                     p["block_tables"], p["max_k"], p["scale"], chunk_keys=chunk)
    torch.cuda.synchronize()
    _assert_same((o1, kc1, vc1), (o2, kc2, vc2), f"H{H_q}/{H_kv} hinted")


def test_suggestion_for_a_skewed_batch_and_its_launch(hip):
    """256 sequences of ~640 keys and one of 8192: the suggestion cuts at 512 keys; the hinted launch meets the
    oracle on the long sequence and on sampled short ones."""
    g = torch.Generator().manual_seed(5)
    ctx = torch.randint(513, 768, (256,), generator=g).tolist() + [8192]
    H_q, H_kv, bs = 16, 8, 256
    chunk = hip.decode_suggest_chunk(len(ctx), H_q, H_kv, max(ctx), sum(ctx))
    assert chunk == 512
    nb = sum(math.ceil(c / bs) for c in ctx) + 2
    p = make_paged_problem(3, [1] * len(ctx), ctx, H_q, H_kv, bs, nb, device=DEV)
    o = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"], p["scale"],
                              chunk_keys=chunk)
    torch.cuda.synchronize()
    rows = [0, 17, 100, 255, 256]
    bt = p["block_tables"].cpu()
    for b in rows:
        ref = ref_attention.paged_attention_ref(p["q"][b:b + 1].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(),
                                                torch.tensor([0, 1], dtype=torch.int32),
                                                torch.tensor([0, ctx[b]], dtype=torch.int32), bt[b:b + 1], p["scale"])
        assert_rows_close(o[b:b + 1], ref, f"sequence {b} ({ctx[b]} keys)")


def test_spread_out_large_batch_gets_the_unsplit_prefetching_launch(hip):
    """384 sequences of 100 .. 2000 keys (more pairs than are resident at once, no sequence that outlasts the
    rest): the suggestion is the plan's cap of 2048 keys, which cuts nothing; the launch meets the oracle on sampled
    rows and the unhinted launch to fp32 summation order."""
    g = torch.Generator().manual_seed(9)
    ctx = sorted(torch.randint(100, 2001, (384,), generator=g).tolist(), reverse=True)
    H_q, H_kv, bs = 16, 8, 256
    assert hip.decode_suggest_chunk(len(ctx), H_q, H_kv, max(ctx), sum(ctx)) == 2048
    nb = sum(math.ceil(c / bs) for c in ctx) + 2
    p = make_paged_problem(4, [1] * len(ctx), ctx, H_q, H_kv, bs, nb, device=DEV)
    plain = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"], p["scale"])
    o = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"], p["scale"],
                              chunk_keys=2048)
    torch.cuda.synchronize()
    assert (o.float() - plain.float()).abs().max().item() <= 2e-2
    bt = p["block_tables"].cpu()
    for b in (0, 1, 77, 200, 383):
        ref = ref_attention.paged_attention_ref(p["q"][b:b + 1].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(),
                                                torch.tensor([0, 1], dtype=torch.int32),
                                                torch.tensor([0, ctx[b]], dtype=torch.int32), bt[b:b + 1], p["scale"])
        assert_rows_close(o[b:b + 1], ref, f"sequence {b} ({ctx[b]} keys)")


def test_spread_out_large_batch_one_launch_step_equals_the_two_launches(hip):
    """The same kind of batch through the one-launch step with the suggested 2048 keys (unsplit, prefetching
    instantiation): every cache byte equal to scatter-then-attention's, outputs equal up to fp32 contraction."""
    g = torch.Generator().manual_seed(10)
    ctx = sorted(torch.randint(100, 2001, (384,), generator=g).tolist(), reverse=True)
    H_q, H_kv = 16, 8
    p, q, k, v, q_w, k_w, cos_sin, pos, slots = _step(22, ctx, H_q, H_kv, 256, True)
    chunk = hip.decode_suggest_chunk(len(ctx), H_q, H_kv, p["max_k"], sum(ctx))
    assert chunk == 2048
    kc1, vc1 = p["k_cache"].clone(), p["v_cache"].clone()
    q_rot, _ = hip.qknorm_rope_store(pos, q, k, v, q_w, k_w, cos_sin, 1e-6, slots, kc1, vc1, want_k=False)
    o1 = hip.attn_decode_paged(q_rot, kc1, vc1, p["cu_k"], p["block_tables"], p["max_k"], p["scale"], chunk_keys=chunk)
    kc2, vc2 = p["k_cache"].clone(), p["v_cache"].clone()
    o2 = hip.attn_decode_paged_fused(pos, q, k, v, q_w, k_w, cos_sin, 1e-6, slots, kc2, vc2, p["cu_k"],
                                     p["block_tables"], p["max_k"], p["scale"], chunk_keys=chunk)
    torch.cuda.synchronize()
    _assert_same((o1, kc1, vc1), (o2, kc2, vc2), "384 spread-out sequences, 2048 keys per work item")


def test_engine_plans_a_skewed_decode_step_with_the_hint():
    """The runner knows every length: on a skewed decode step its context carries the suggested chunk, the step
    captured as a graph is keyed by it, and graph replay and eager step give the same logits."""
    import nanovllm_amd  # noqa: F401
    from nanovllm_amd.engine.llm_engine import LLMEngine
    from nanovllm_amd.engine.model_runner import prepare_inputs
    from nanovllm_amd.engine.sequence import SamplingParams
    from nanovllm_amd.model.qwen3 import ModelConfig
    from nanovllm_amd.utils.cache_context import get_context, set_context
    from test_gpu_engine import build

    cfg = ModelConfig(vocab_size=2048, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096)
    model = build(cfg, None, True)
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=300, max_num_batched_tokens=8192,
                    long_prefill_token_threshold=512, device=DEV, eos=-1, use_graphs=True)
    g = torch.Generator().manual_seed(1)
    prompts = [torch.randint(0, cfg.vocab_size, (int(n),), generator=g).tolist()
               for n in torch.randint(20, 60, (256,), generator=g)]
    prompts.append(torch.randint(0, cfg.vocab_size, (3000,), generator=g).tolist())
    for pr in prompts:
        eng.add_request(pr, SamplingParams(0.0, 16))
    runner, sched = eng.runner, eng.scheduler
    # run until a pure decode step over all 257 sequences is scheduled
    for _ in range(64):
        batch, _ = sched.schedule()
        if len(batch) == len(prompts) and all(s.num_scheduled_tokens == 1 and s.output_tokens for s in batch):
            break
        sched.update_from_output(batch, runner.run(batch))
    else:
        pytest.fail("no full decode step reached")
    m = prepare_inputs(batch, runner.block_size)
    graph = runner._graph_for(len(batch), m["block_tables"].shape[1], m)
    assert graph.chunk_keys > 0 and graph.ctx.decode_chunk_keys == graph.chunk_keys
    set_context(graph.ctx)
    with torch.inference_mode():
        logits_g = graph.replay(m).float().clone()
        ids, pos, m2 = runner.set_step_context(batch)
        assert get_context().decode_chunk_keys > 0
        last = torch.from_numpy(m2["cu_seqlens_q"][1:].astype("int64") - 1).to(DEV)
        logits_e = runner.model(ids, pos, logits_rows=last).float()
    torch.cuda.synchronize()
    assert torch.isfinite(logits_g).all()
    assert (logits_g - logits_e).abs().max().item() <= 2e-2
    # the rest of the run replays captured graphs whose chunk was chosen for other lengths of the same skew
    seqs_all = list(batch)
    for _ in range(64):
        sched.update_from_output(batch, runner.run(batch))
        if sched.is_finished():
            break
        batch, _ = sched.schedule()
    assert sched.is_finished() and all(len(s.output_tokens) == 16 for s in seqs_all)
    assert any(k[2] > 0 for k in runner._graphs)
    # a balanced batch of the same size keeps the plan's own choice
    assert runner._decode_chunk(dict(num_leading_decode=256, sum_seqlen_k_decode=256 * 640), 768) == 0


def test_small_launches_combine_in_launch_on_the_zeroed_buffer_and_leave_it_zero(hip):
    """Launches with at most 8 (sequence, kv head) pairs go through ops' zero-initialised buffer and combine their
    key ranges inside the launch (NV_DECODE_WS_COUNTERS_ZERO): several launches in a row, of different sizes, meet
    the oracle, and the arrival counters are zero again after each."""
    from nanovllm_amd import ops
    for i, (H_q, H_kv, ctx) in enumerate([(16, 8, [9000]), (8, 1, [3000, 70, 5000]), (32, 8, [20000]), (16, 8, [300])]):
        bs = 256
        nb = sum(math.ceil(c / bs) for c in ctx) + 2
        p = make_paged_problem(70 + i, [1] * len(ctx), ctx, H_q, H_kv, bs, nb, device=DEV)
        ref = ref_attention.paged_attention_ref(p["q"].cpu(), p["k_cache"].cpu(), p["v_cache"].cpu(), p["cu_q"].cpu(),
                                                p["cu_k"].cpu(), p["block_tables"].cpu(), p["scale"])
        o = hip.attn_decode_paged(p["q"], p["k_cache"], p["v_cache"], p["cu_k"], p["block_tables"], p["max_k"], p["scale"])
        torch.cuda.synchronize()
        assert_rows_close(o, ref, f"small launch {i}")
        assert ops._small_decode_ws, "the small launches' buffer was not used"
        for ws in ops._small_decode_ws.values():
            assert int(ws[:65536].view(torch.int32).abs().sum()) == 0, "arrival counters not back at zero"
