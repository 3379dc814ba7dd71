"""CPU, world_size 2 over gloo: head-sharded attention + all-gather reproduces the unsharded result
bit for bit (the oracle stands in for the per-rank attention kernel: sharding must not change the
per-head arithmetic), and the gathered heads come back in global order."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import make_paged_problem
from nanovllm_amd.parallel import HeadShard, all_gather_heads
from oracle import ref_attention


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, H_q, H_kv, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = make_paged_problem(123, [1, 37, 1, 5], [70, 37, 300, 64], H_q, H_kv, 64, 12)
        shard = HeadShard(rank, world, H_q, H_kv)
        o_local = ref_attention.paged_attention_ref(shard.slice_q(p["q"]).contiguous(),
                                                    shard.slice_kv(p["k_cache"]).contiguous(),
                                                    shard.slice_kv(p["v_cache"]).contiguous(),
                                                    p["cu_q"], p["cu_k"], p["block_tables"], p["scale"]).bfloat16()
        o = all_gather_heads(o_local, shard)
        full = ref_attention.paged_attention_ref(p["q"], p["k_cache"], p["v_cache"], p["cu_q"], p["cu_k"],
                                                 p["block_tables"], p["scale"]).bfloat16()
        ok = torch.equal(o.view(torch.int16), full.view(torch.int16))
        # ordering check with a tensor that encodes (rank, head)
        tag = torch.full((3, shard.local_heads, 128), 0.0)
        for i, h in enumerate(shard.q_heads):
            tag[:, i] = float(h)
        g = all_gather_heads(tag, shard)
        ok = ok and all(bool((g[:, h] == float(h)).all()) for h in range(H_q))
        # persistent buffers: a step with more tokens than any before makes them grow, and the outgrown
        # pair stays alive (decode graphs captured earlier replay their gather at the old addresses)
        from nanovllm_amd.parallel import HeadGather
        hg = HeadGather(shard)
        small = hg(tag)
        old_staging, old_out = hg._staging, hg._out
        big = hg(tag.repeat(5, 1, 1))
        ok = ok and hg._staging is not old_staging and any(t is old_staging for t in hg._retired) \
            and any(t is old_out for t in hg._retired)
        ok = ok and all(bool((big[:, h] == float(h)).all()) for h in range(H_q)) and small.shape[0] == 3
        again = hg(tag)                                   # a smaller step reuses the grown buffers
        ok = ok and hg._staging.shape[0] == 5 * 3 * world and len(hg._retired) == 2 and bool((again == g).all())
        flags = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([1.0 if ok else 0.0]))
        if rank == 0:
            with open(result_path, "w") as f:
                f.write("ok" if all(x.item() == 1.0 for x in flags) else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("H_q,H_kv", [(8, 2), (16, 8)])
def test_head_sharded_attention_all_gather_world2(tmp_path, H_q, H_kv):
    path = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(2, _free_port(), H_q, H_kv, path), nprocs=2, join=True)
    assert open(path).read() == "ok"


def _engine_worker(rank, world, port, result_path):
    """Every rank runs the same scheduler on the same requests (replicated control plane) with a
    head-sharded model; the generated tokens must equal those of the unsharded model."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nanovllm_amd.engine.llm_engine import LLMEngine
        from nanovllm_amd.engine.sequence import SamplingParams
        from nanovllm_amd.model.qwen3 import ModelConfig, Qwen3ForCausalLM
        from oracle import ref_modules
        cfg = ModelConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                          num_attention_heads=8, num_key_value_heads=4, head_dim=128, max_position_embeddings=1024)
        torch.manual_seed(7)
        full = Qwen3ForCausalLM(cfg, layers=ref_modules, fused=False).to(torch.bfloat16)
        for prm in full.parameters():
            torch.nn.init.normal_(prm, 0.0, 0.05)
        shard = HeadShard(rank, world, cfg.num_attention_heads, cfg.num_key_value_heads)
        part = Qwen3ForCausalLM(cfg, layers=ref_modules, fused=False, shard=shard).to(torch.bfloat16)
        part.load_full_state_dict(full.state_dict())
        g = torch.Generator().manual_seed(11)
        prompts = [torch.randint(0, 512, (n,), generator=g).tolist() for n in (16, 45, 70)]
        kw = dict(block_size=16, num_kvcache_blocks=32, max_num_batched_tokens=64, long_prefill_token_threshold=32,
                  device="cpu", eos=-1)
        eng = LLMEngine(part, cfg, **kw)
        ok = eng.runner.kv_cache.shape[4] == cfg.num_key_value_heads // world
        outs, _ = eng.generate(prompts, SamplingParams(0.0, 5))
        ref_outs, _ = LLMEngine(full, cfg, **kw).generate(prompts, SamplingParams(0.0, 5))
        ok = ok and [o["token_ids"] for o in outs] == [o["token_ids"] for o in ref_outs]
        # the all-gather buffers persist: one pair for all layers, no allocation per layer or per step
        gathers = {id(layer.self_attn._gather) for layer in part.model.layers}
        buf = part.model.layers[0].self_attn._gather._staging
        ok = ok and len(gathers) == 1 and buf is not None
        # temperature > 0 with DIFFERENT generator states per rank: rank 0's draws are broadcast, so the
        # ranks keep scheduling identical batches (a mismatch would hang or corrupt the next all-gather)
        torch.manual_seed(1000 + rank)
        sampled, _ = LLMEngine(part, cfg, **kw).generate(prompts, SamplingParams(0.8, 6))
        ok = ok and part.model.layers[0].self_attn._gather._staging is buf
        mine = torch.tensor([o["token_ids"] for o in sampled])
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        ok = ok and all(torch.equal(b, both[0]) for b in both)
        flags = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([1.0 if ok else 0.0]))
        if rank == 0:
            with open(result_path, "w") as f:
                f.write("ok" if all(x.item() == 1.0 for x in flags) else "mismatch")
    finally:
        dist.destroy_process_group()


def test_head_sharded_model_generates_same_tokens_world2(tmp_path):
    path = str(tmp_path / "result.txt")
    mp.spawn(_engine_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    assert open(path).read() == "ok"


def test_head_shard_ranges():
    s = HeadShard(3, 4, 64, 8)
    assert list(s.kv_heads) == [6, 7] and list(s.q_heads) == list(range(48, 64))
    assert s.local_heads == 16 and s.local_kv_heads == 2
    with pytest.raises(ValueError):
        HeadShard(0, 3, 64, 8)
    one = HeadShard(0, 1, 16, 8)
    x = torch.zeros(2, 16, 128)
    assert all_gather_heads(x, one) is x


def _bench_reduce_worker(rank, world, port, result_path):
    """bench.py's cross-rank arithmetic on gloo: the step time is the MAX over the ranks, the bytes their SUM,
    and the per-rank device records come back in rank order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        t = bench.reduce_scalar(dist, torch.device("cpu"), 1.0 + rank, "max")
        b = bench.reduce_scalar(dist, torch.device("cpu"), 100.0 * (rank + 1), "sum")
        ids = bench._gather_objects(dist, world, {"rank": rank})
        ok = t == float(world) and b == 100.0 * world * (world + 1) / 2 and [d["rank"] for d in ids] == list(range(world))
        alone = bench.reduce_scalar(None, torch.device("cpu"), 3.5, "max") == 3.5 and \
            bench._gather_objects(None, 1, "x") == ["x"]
        flags = [torch.zeros(1) for _ in range(world)]
        dist.all_gather(flags, torch.tensor([1.0 if ok and alone else 0.0]))
        if rank == 0:
            with open(result_path, "w") as f:
                f.write("ok" if all(x.item() == 1.0 for x in flags) else "mismatch")
    finally:
        dist.destroy_process_group()


def test_bench_aggregates_over_ranks_world2(tmp_path):
    result = tmp_path / "result.txt"
    mp.spawn(_bench_reduce_worker, args=(2, _free_port(), str(result)), nprocs=2, join=True)
    assert result.read_text() == "ok"
