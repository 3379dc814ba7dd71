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
        max_num_batched_tokens=4000, fused=True, graphs=True, fuse_gemms=True):
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
                    device=device, eos=-1, use_graphs=graphs)
    g = torch.Generator().manual_seed(seed)
    prompts = [torch.randint(0, cfg.vocab_size, (prompt_len,), generator=g).tolist() for _ in range(batch)]
    # warm-up on a small batch (library load, hipBLASLt heuristics)
    eng.generate(prompts[:4], SamplingParams(0.0, 4))
    eng.generate(prompts, SamplingParams(0.0, 2))          # captures the decode graph of the full batch
    torch.cuda.synchronize()
    outs, metrics = eng.generate(prompts, SamplingParams(0.0, out_len))
    s = metrics.summary()
    s.update(model=model_name, batch=batch, prompt_len=prompt_len, out_len=out_len, fused=fused, graphs=graphs, fuse_gemms=fuse_gemms)
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
    a = ap.parse_args()
    print(json.dumps(run(a.model, a.batch, a.prompt_len, a.out_len, fused=not a.unfused, graphs=not a.no_graphs, fuse_gemms=not a.no_fuse_gemms)))
