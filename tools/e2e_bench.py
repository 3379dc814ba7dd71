#!/usr/bin/env python3
"""End-to-end generation throughput with the HIP hot path: BASELINE config 2 by default
(Qwen3-0.6B shapes, random-init weights, 256 prompts x 512 random tokens, 256 output tokens each,
argmax decoding, EOS disabled, KV block 256)."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nanovllm_amd  # noqa: E402,F401
from nanovllm_amd.engine.llm_engine import LLMEngine  # noqa: E402
from nanovllm_amd.engine.sequence import SamplingParams  # noqa: E402
from nanovllm_amd.model.qwen3 import ModelConfig, Qwen3ForCausalLM  # noqa: E402


def run(model_name="qwen3_0_6b", batch=256, prompt_len=512, out_len=256, device="cuda:0", seed=0,
        max_num_batched_tokens=4000, fused=True, graphs=True, fuse_gemms=True, pipelined=True):
    cfg = getattr(ModelConfig, model_name)()
    torch.manual_seed(seed)
    with torch.device(device):
        model = Qwen3ForCausalLM(cfg, fused=fused)
    for name, p in model.named_parameters():
        if "norm" in name:
            torch.nn.init.ones_(p)
        else:
            torch.nn.init.normal_(p, 0.0, 0.02)
    model = model.to(torch.bfloat16)
    if fuse_gemms:
        model.fuse_weights()
    blocks = batch * (-(-(prompt_len + out_len) // 256)) + 8
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=blocks, max_num_batched_tokens=max_num_batched_tokens,
                    device=device, eos=-1, use_graphs=graphs, pipelined_decode=pipelined)
    g = torch.Generator().manual_seed(seed)
    prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    # warm-up (library load, hipBLASLt heuristics, decode-graph capture of the full batch) on OTHER random
    # prompts, so that the timed run finds nothing of its own in the prefix cache
    warm = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    eng.generate(warm[:4], SamplingParams(0.0, 4))
    eng.generate(warm, SamplingParams(0.0, 2))
    torch.cuda.synchronize()
    outs, metrics = eng.generate(prompts, SamplingParams(0.0, out_len))
    s = metrics.summary()
    s["prefix_cache_hit_tokens"] = int(sum(o.get("cached_tokens", 0) for o in outs))
    s.update(model=model_name, batch=batch, prompt_len=prompt_len, out_len=out_len, fused=fused, graphs=graphs, fuse_gemms=fuse_gemms,
             pipelined_decode=pipelined)
    return s


def _build(model_name, device, seed, fused=True, fuse_gemms=True):
    cfg = getattr(ModelConfig, model_name)()
    torch.manual_seed(seed)
    with torch.device(device):
        model = Qwen3ForCausalLM(cfg, fused=fused)
    for name, p in model.named_parameters():
        if "norm" in name:
            torch.nn.init.ones_(p)
        else:
            torch.nn.init.normal_(p, 0.0, 0.02)
    model = model.to(torch.bfloat16)
    return cfg, (model.fuse_weights() if fuse_gemms else model)


def run_long_prompt_with_shared_prefix(model_name="qwen3_8b", prompt_len=8192, shared=7936, tail=256, out_len=16,
                                       device="cuda:0", seed=0):
    """One long prompt prefilled in 512-token chunks, then a second request that shares its first
    `shared` tokens (full cache blocks) and adds `tail` new ones: the second must skip the shared part."""
    cfg, model = _build(model_name, device, seed)
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=2 * (prompt_len // 256) + 16, device=device, eos=-1,
                    use_graphs=False)
    g = torch.Generator().manual_seed(seed)
    warm = torch.randint(0, cfg.vocab_size, (1024,), generator=g).tolist()
    eng.generate([warm], SamplingParams(0.0, 4))
    p1 = torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist()
    p2 = p1[:shared] + torch.randint(0, cfg.vocab_size, (tail,), generator=g).tolist()
    torch.cuda.synchronize()
    o1, m1 = eng.generate([p1], SamplingParams(0.0, out_len))
    o2, m2 = eng.generate([p2], SamplingParams(0.0, out_len))
    s1, s2 = m1.summary(), m2.summary()
    return {"model": model_name, "prompt_len": prompt_len, "chunk": 512,
            "first": {"ttft_s": s1["ttft_max_s"], "prefill_tok_per_s": prompt_len / s1["ttft_max_s"],
                      "cached_tokens": o1[0]["cached_tokens"], "mean_tpot_ms": s1["mean_tpot_ms"]},
            "second_shared_prefix": {"ttft_s": s2["ttft_max_s"], "cached_tokens": o2[0]["cached_tokens"],
                                     "new_prompt_tokens": len(p2) - o2[0]["cached_tokens"],
                                     "mean_tpot_ms": s2["mean_tpot_ms"]}}


def run_long_context_decode(model_name="llama31_8b", batch=8, ctx=32768, out_len=64, device="cuda:0", seed=0):
    """`batch` sequences whose context reaches `ctx` tokens at the last generated token: prefill in
    512-token chunks, then `out_len` decode steps at ~32k keys per sequence."""
    cfg, model = _build(model_name, device, seed)
    prompt_len = ctx - out_len
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=batch * (ctx // 256 + 1) + 8, device=device,
                    eos=-1, use_graphs=True, pipelined_decode=True)
    g = torch.Generator().manual_seed(seed)
    warm = [torch.randint(0, cfg.vocab_size, (600,), generator=g).tolist() for _ in range(batch)]
    eng.generate(warm, SamplingParams(0.0, 3))
    prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    torch.cuda.synchronize()
    outs, m = eng.generate(prompts, SamplingParams(0.0, out_len))
    s = m.summary()
    s.update(model=model_name, batch=batch, ctx=ctx, out_len=out_len,
             prefill_tok_per_s=batch * prompt_len / s["ttft_max_s"],
             prefix_cache_hit_tokens=int(sum(o["cached_tokens"] for o in outs)))
    return s


def run_skewed_batch(model_name="qwen3_0_6b", n_short=255, short_len=512, long_len=16384, out_len=128, device="cuda:0",
                     seed=0, hints=True):
    """`n_short` prompts of `short_len` tokens and ONE of `long_len`, all decoding together: the decode steps'
    attention launch has one sequence 30 x as long as the others.  `hints`: the runner plans the launch with
    the step's lengths (default) or from the longest one alone."""
    cfg, model = _build(model_name, device, seed)
    blocks = n_short * (-(-(short_len + out_len) // 256)) + (long_len + out_len) // 256 + 16
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=blocks, device=device, eos=-1, use_graphs=True,
                    pipelined_decode=True)
    eng.runner.use_length_hints = hints
    g = torch.Generator().manual_seed(seed)
    warm = [torch.randint(0, cfg.vocab_size, (short_len,), generator=g).tolist() for _ in range(8)]
    eng.generate(warm, SamplingParams(0.0, 3))
    prompts = [torch.randint(0, cfg.vocab_size, (long_len,), generator=g).tolist()]
    prompts += [torch.randint(0, cfg.vocab_size, (short_len,), generator=g).tolist() for _ in range(n_short)]
    torch.cuda.synchronize()
    outs, m = eng.generate(prompts, SamplingParams(0.0, out_len))
    s = m.summary()
    s.update(model=model_name, n_short=n_short, short_len=short_len, long_len=long_len, out_len=out_len,
             length_hints=hints, decode_graphs=sorted(eng.runner._graphs))
    return s


def run_uneven_batch(model_name="qwen3_0_6b", batch=512, lo=100, hi=2000, out_len=64, device="cuda:0", seed=0,
                     longest_first=True):
    """`batch` prompts of U[lo, hi] tokens decoding together: more decode work items than the chip holds at once,
    of very different lengths.  `longest_first`: the runner hands decode rows to the launch longest first
    (default) or in the scheduler's order."""
    from nanovllm_amd.engine.model_runner import ModelRunner
    if not longest_first:
        ModelRunner.sort_decode_rows_from = 1 << 30
    cfg, model = _build(model_name, device, seed)
    g = torch.Generator().manual_seed(seed)
    lens = torch.randint(lo, hi + 1, (batch,), generator=g).tolist()
    blocks = sum(-(-(n + out_len) // 256) for n in lens) + 16
    eng = LLMEngine(model, cfg, block_size=256, num_kvcache_blocks=blocks, max_num_batched_tokens=16384, device=device,
                    eos=-1, use_graphs=True, pipelined_decode=True)
    warm = [torch.randint(0, cfg.vocab_size, (300,), generator=g).tolist() for _ in range(8)]
    eng.generate(warm, SamplingParams(0.0, 3))
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in lens]
    torch.cuda.synchronize()
    outs, m = eng.generate(prompts, SamplingParams(0.0, out_len))
    s = m.summary()
    s.update(model=model_name, batch=batch, prompt_lens=f"U[{lo},{hi}]", out_len=out_len, longest_first=longest_first)
    return s


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="qwen3_0_6b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--prompt-len", type=int, default=512)
    ap.add_argument("--out-len", type=int, default=256)
    ap.add_argument("--unfused", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-fuse-gemms", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="synchronous decode loop (token values read every step)")
    ap.add_argument("--scenario", default="batch", choices=["batch", "long_prompt", "long_context_decode", "skewed_batch", "uneven_batch"])
    ap.add_argument("--no-length-hints", action="store_true", help="skewed_batch: plan decode launches from the longest length alone")
    ap.add_argument("--scheduler-order", action="store_true", help="uneven_batch: decode rows stay in the scheduler's order")
    ap.add_argument("--ctx", type=int, default=32768)
    ap.add_argument("--prefill-kernel", type=int, default=-1, choices=[-1, 0, 1],
                    help="force the compiler-scheduled (0) or the hand-scheduled (1) prefill kernel; -1 = heuristic")
    a = ap.parse_args()
    if a.prefill_kernel >= 0:
        from nanovllm_amd import _lib
        _lib.set_tuning(prefill_kernel=a.prefill_kernel)
    if a.scenario == "long_prompt":
        print(json.dumps(run_long_prompt_with_shared_prefix(a.model if a.model != "qwen3_0_6b" else "qwen3_8b")))
        sys.exit(0)
    if a.scenario == "skewed_batch":
        print(json.dumps(run_skewed_batch(a.model, hints=not a.no_length_hints)))
        sys.exit(0)
    if a.scenario == "uneven_batch":
        print(json.dumps(run_uneven_batch(a.model, a.batch if a.batch != 256 else 512, out_len=a.out_len if a.out_len != 256 else 64,
                                          longest_first=not a.scheduler_order)))
        sys.exit(0)
    if a.scenario == "long_context_decode":
        print(json.dumps(run_long_context_decode(a.model if a.model != "qwen3_0_6b" else "llama31_8b",
                                                 a.batch if a.batch != 256 else 8, a.ctx, a.out_len if a.out_len != 256 else 64)))
        sys.exit(0)
    print(json.dumps(run(a.model, a.batch, a.prompt_len, a.out_len, fused=not a.unfused, graphs=not a.no_graphs, fuse_gemms=not a.no_fuse_gemms,
                         pipelined=not a.no_pipeline)))
