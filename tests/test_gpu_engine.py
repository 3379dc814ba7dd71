"""GPU: the engine end to end with the HIP layers -- greedy tokens identical to the same engine
running the oracle's modules (eager torch, fp32 attention) on the same device, so both share the
GEMMs and differ only in the attention hot path."""
import pytest
import torch

import nanovllm_amd  # noqa: F401
from nanovllm_amd.engine.llm_engine import LLMEngine
from nanovllm_amd.engine.sequence import SamplingParams
from nanovllm_amd.model.qwen3 import ModelConfig, Qwen3ForCausalLM
from oracle import ref_modules

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(cfg, layers, fused, seed=0):
    torch.manual_seed(seed)
    model = Qwen3ForCausalLM(cfg, layers=layers, fused=fused)
    for name, p in model.named_parameters():
        if "norm" in name:
            torch.nn.init.normal_(p, 1.0, 0.05)
        else:
            torch.nn.init.normal_(p, 0.0, 0.02)
    return model.to(torch.bfloat16).to(DEV)


def _drive(engine, prompts, max_tokens, forced=None):
    """Run the engine step by step with argmax decoding.  Returns per sequence the list of
    (chosen token, logits row) for every sampled position.  With `forced` (token lists per
    sequence) the engine is teacher-forced: the given token is appended instead of its own argmax."""
    seqs = [engine.add_request(p, SamplingParams(0.0, max_tokens)) for p in prompts]
    index = {s.id: i for i, s in enumerate(seqs)}
    trace = [[] for _ in seqs]
    runner, sched = engine.runner, engine.scheduler
    while not sched.is_finished():
        batch, _ = sched.schedule()
        ids, pos, meta = runner.set_step_context(batch)
        last = torch.from_numpy(meta["cu_seqlens_q"][1:].astype("int64") - 1).to(DEV)
        with torch.inference_mode():
            logits = runner.model(ids, pos, logits_rows=last).float()
        picks = logits.argmax(dim=-1).tolist()
        out = []
        for row, (s, tok) in enumerate(zip(batch, picks)):
            if s.num_computed_tokens + s.num_scheduled_tokens < len(s.prompt_tokens):
                out.append(None)
                continue
            i = index[s.id]
            if forced is not None:
                tok = forced[i][len(trace[i])]
            trace[i].append((tok, logits[row].cpu()))
            out.append(tok)
        sched.update_from_output(batch, out)
    return trace


@pytest.mark.parametrize("fused,graphs", [(True, False), (False, False), (True, True)])
def test_greedy_tokens_over_128_steps(fused, graphs):
    """Qwen3-0.6B-shaped attention (16 q / 8 kv heads x 128, q/k-norm, rope theta 1e6, block 256),
    fewer / narrower layers and a 4k vocabulary to keep the oracle run short; 4 prompts of lengths
    16/13/512/502, argmax decoding for 128 steps through chunked prefill and batched decode.

    The HIP engine runs freely.  The oracle-module engine is then teacher-forced with the HIP
    engine's tokens, which gives the reference logits for exactly the same contexts at every step.
    Greedy tokens must be identical wherever the reference's top-2 margin is larger than the
    numerical tolerance of the path; inside that margin (a numerical tie of random-init logits)
    the HIP token must be one of the tied candidates."""
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (16, 13, 512, 502)]
    hip_model = build(cfg, None, fused)
    ref_model = build(cfg, ref_modules, False)
    ref_model.load_state_dict(hip_model.state_dict())
    kw = dict(block_size=256, num_kvcache_blocks=16, max_num_batched_tokens=600, long_prefill_token_threshold=512,
              device=DEV, eos=-1)
    hip = LLMEngine(hip_model, cfg, use_graphs=graphs, **kw)
    ref = LLMEngine(ref_model, cfg, **kw)
    if graphs:
        hip_model.fuse_weights()                  # fused q/k/v and gate/up GEMMs: same math, fewer launches
    if graphs:      # the public generate() path exercises graph replay; compare its tokens with the eager HIP run
        outs, m = hip.generate(prompts, SamplingParams(0.0, 128))
        eager = LLMEngine(hip_model, cfg, **kw)
        outs_eager, _ = eager.generate(prompts, SamplingParams(0.0, 128))
        assert [o["token_ids"] for o in outs] == [o["token_ids"] for o in outs_eager]
        assert m.summary()["generated_tokens"] == 4 * 128 and len(hip.runner._graphs) >= 1
        return
    hip_trace = _drive(hip, prompts, 128)
    hip_tokens = [[t for t, _ in tr] for tr in hip_trace]
    ref_trace = _drive(ref, prompts, 128, forced=hip_tokens)
    assert all(len(t) == 128 for t in hip_tokens)
    ties = 0
    for tr_h, tr_r in zip(hip_trace, ref_trace):
        for (tok, lg_h), (_, lg_r) in zip(tr_h, tr_r):
            scale = max(1.0, lg_r.abs().max().item())
            assert (lg_h - lg_r).abs().max().item() <= 2e-2 * scale           # logits agree at every step
            top2 = lg_r.topk(2).values
            if tok != int(lg_r.argmax()):
                ties += 1
                assert (top2[0] - lg_r[tok]).item() <= 2e-2 * scale, "token differs outside a numerical tie"
    assert ties <= 8, f"{ties} of 512 greedy tokens fell on numerical ties"
    assert hip.scheduler.block_manager.num_free_blocks == 16


def test_logits_close_on_a_mixed_step():
    """One engine step with a prefill chunk and decode rows: logits of HIP vs oracle modules."""
    cfg = ModelConfig(vocab_size=2048, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048)
    g = torch.Generator().manual_seed(3)
    hip_model = build(cfg, None, True, seed=1)
    ref_model = build(cfg, ref_modules, False, seed=1)
    ref_model.load_state_dict(hip_model.state_dict())
    outs = []
    for model in (hip_model, ref_model):
        eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=8, max_num_batched_tokens=4000, device=DEV, eos=-1)
        gg = torch.Generator().manual_seed(4)
        s1 = eng.add_request(torch.randint(0, 2048, (300,), generator=gg).tolist(), SamplingParams(0.0, 4))
        eng.step()
        eng.step()                                           # s1 now decodes
        eng.add_request(torch.randint(0, 2048, (200,), generator=gg).tolist(), SamplingParams(0.0, 4))
        seqs, _ = eng.scheduler.schedule()                  # mixed: s1 decode row + new prefill
        ids, pos, meta = eng.runner.set_step_context(seqs)
        with torch.inference_mode():
            outs.append(model(ids, pos).float().cpu())
        assert meta["max_seqlen_q"] == 200 and meta["num_decode_seqs"] == 1
    err = (outs[0] - outs[1]).abs().max().item()
    scale = outs[1].abs().max().item()
    assert err <= 2e-2 * max(1.0, scale), (err, scale)


@pytest.mark.parametrize("graphs", [False, True])
def test_pipelined_decode_matches_synchronous_loop(graphs):
    """Decode steps enqueued one ahead of the host (input tokens gathered on the device from the
    previous step's samples) give exactly the tokens of the synchronous loop on the same kernels."""
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096)
    model = build(cfg, None, True, seed=3)
    g = torch.Generator().manual_seed(8)
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (16, 13, 300, 255, 257, 40)]
    params = [SamplingParams(0.0, n) for n in (40, 3, 64, 1, 33, 64)]
    kw = dict(block_size=256, num_kvcache_blocks=24, max_num_batched_tokens=400, long_prefill_token_threshold=256,
              device=DEV, eos=-1, use_graphs=graphs)
    ref, _ = LLMEngine(model, cfg, **kw).generate(prompts, params)
    out, m = LLMEngine(model, cfg, pipelined_decode=True, **kw).generate(prompts, params)
    assert [o["token_ids"] for o in out] == [o["token_ids"] for o in ref]
    assert m.num_generated_tokens == sum(p.max_tokens for p in params)


def test_prefix_cache_hit_gives_the_same_tokens_on_the_hip_path():
    """A second request sharing two full cache blocks with an earlier one skips them (chunked prefill
    with keys already cached: the bottom-right mask of the prefill kernel) and must produce the tokens
    of an engine that computes the whole prompt."""
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096)
    model = build(cfg, None, True, seed=5)
    g = torch.Generator().manual_seed(12)
    shared = torch.randint(0, cfg.vocab_size, (512,), generator=g).tolist()
    p1 = shared + torch.randint(0, cfg.vocab_size, (40,), generator=g).tolist()
    p2 = shared + torch.randint(0, cfg.vocab_size, (75,), generator=g).tolist()
    kw = dict(block_size=256, num_kvcache_blocks=16, max_num_batched_tokens=400, long_prefill_token_threshold=256,
              device=DEV, eos=-1, use_graphs=True, pipelined_decode=True)
    eng = LLMEngine(model, cfg, **kw)
    eng.generate([p1], SamplingParams(0.0, 8))
    out, _ = eng.generate([p2], SamplingParams(0.0, 24))
    assert out[0]["cached_tokens"] == 512
    fresh, _ = LLMEngine(model, cfg, **kw).generate([p2], SamplingParams(0.0, 24))
    assert fresh[0]["cached_tokens"] == 0
    assert out[0]["token_ids"] == fresh[0]["token_ids"]


@pytest.mark.parametrize("graphs", [False, True])
def test_preemption_on_the_hip_path(graphs):
    """Requests that together outgrow the KV cache, on the HIP kernels with pipelined decode (and captured decode
    graphs): the youngest running sequence is preempted and its K/V recomputed later by the prefill kernel.  Every
    request finishes with its full token count, all blocks come back, and every request's first token (sampled
    before any preemption) is the big-cache run's.  Later tokens are not compared: the two runs batch differently
    (other launch plans and GEMM shapes, i.e. other fp32 summation orders), and a random-init model's greedy
    tokens flip at numerical ties; exact token equality under preemption is shown on the CPU with the fp32 oracle
    layers (tests/test_engine_cpu.py)."""
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096)
    model = build(cfg, None, True, seed=9)
    g = torch.Generator().manual_seed(21)
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (300, 280, 100, 200)]
    params = [SamplingParams(0.0, n) for n in (260, 50, 200, 120)]
    kw = dict(block_size=256, max_num_batched_tokens=600, long_prefill_token_threshold=256, device=DEV, eos=-1,
              use_graphs=graphs, pipelined_decode=True)
    want, _ = LLMEngine(model, cfg, num_kvcache_blocks=16, **kw).generate(prompts, params)
    assert all(o["num_preemptions"] == 0 for o in want)
    small = LLMEngine(model, cfg, num_kvcache_blocks=5, **kw)      # 1280 tokens of cache for 1510 tokens of requests
    got, m = small.generate(prompts, params)
    assert [len(o["token_ids"]) for o in got] == [p.max_tokens for p in params]
    assert m.num_generated_tokens == sum(p.max_tokens for p in params)
    assert sum(o["num_preemptions"] for o in got) >= 1
    assert small.scheduler.block_manager.num_free_blocks == 5
    assert [o["token_ids"][0] for o in got] == [w["token_ids"][0] for w in want]
