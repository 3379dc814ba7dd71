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


@pytest.mark.parametrize("fused", [True, False])
def test_greedy_tokens_identical_over_128_steps(fused):
    """Qwen3-0.6B-shaped attention (16 q / 8 kv heads x 128, q/k-norm, rope theta 1e6, block 256),
    fewer / narrower layers and a 4k vocabulary to keep the oracle run short; 4 prompts of lengths
    16/13/512/502, argmax decoding for 128 steps through chunked prefill and batched decode."""
    cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=4,
                      num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=2048)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (16, 13, 512, 502)]
    sp = SamplingParams(0.0, 128)
    hip_model = build(cfg, None, fused)
    ref_model = build(cfg, ref_modules, False)
    ref_model.load_state_dict(hip_model.state_dict())
    hip = LLMEngine(hip_model, cfg, block_size=256, num_kvcache_blocks=16, max_num_batched_tokens=600,
                    long_prefill_token_threshold=512, device=DEV, eos=-1)
    ref = LLMEngine(ref_model, cfg, block_size=256, num_kvcache_blocks=16, max_num_batched_tokens=600,
                    long_prefill_token_threshold=512, device=DEV, eos=-1)
    out_hip, m = hip.generate(prompts, sp)
    out_ref, _ = ref.generate(prompts, sp)
    for a, b in zip(out_hip, out_ref):
        assert len(a["token_ids"]) == 128
        assert a["token_ids"] == b["token_ids"]
    assert m.summary()["generated_tokens"] == 4 * 128
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
