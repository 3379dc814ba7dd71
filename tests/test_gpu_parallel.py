"""GPU, needs >= 2 GPUs (skipped on a one-GPU box): head-sharded decode and prefill attention on
BASELINE config 4 shapes (Qwen3-32B: 64 query / 8 kv heads) with one rank per GPU over RCCL.  Sharding by
kv head leaves the per-head arithmetic unchanged, so the all-gathered prefill outputs must be bit-identical
to the single-GPU launch; decode launches plan their key split from the number of kv heads per GPU, so their
outputs may differ in the last bits of a different fp32 summation order and are held to 1e-2.
A second test runs a head-sharded ENGINE with captured decode graphs (the per-layer RCCL all-gather inside
the HIP graph), greedy and temperature > 0, against an unsharded engine on rank 0."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu
N_GPUS = torch.cuda.device_count()          # does not initialise the GPU


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, result_path):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from conftest import make_paged_problem
        from nanovllm_amd import ops
        from nanovllm_amd.parallel import HeadGather, HeadShard
        H_q, H_kv, bs = 64, 8, 256
        shard = HeadShard(rank, world, H_q, H_kv)
        gather = HeadGather(shard)
        ok = True
        for n_qs, n_ks in (([1] * 24, [2048 - 7 * i for i in range(24)]),          # decode, ctx ~2048
                           ([300, 512, 1, 64], [300, 1024, 700, 64])):               # prefill / chunked / mixed
            nb = sum((n + bs - 1) // bs for n in n_ks) + 1
            p = make_paged_problem(17, n_qs, n_ks, H_q, H_kv, bs, nb, device=dev)      # same seed on every rank
            q_l = shard.slice_q(p["q"])                                                # strided view: token stride kept
            kc, vc = shard.slice_kv(p["k_cache"]).contiguous(), shard.slice_kv(p["v_cache"]).contiguous()
            if max(n_qs) == 1:
                run = lambda q, k, v: ops.attn_decode_paged(q, k, v, p["cu_k"], p["block_tables"], p["max_k"], p["scale"],
                                                            cu_seqlens_q=p["cu_q"])
            else:
                def run(q, k, v):
                    o = ops.attn_prefill_varlen(q, k, v, p["cu_q"], p["cu_k"], p["block_tables"], p["max_q"], p["max_k"],
                                                p["scale"], skip_single_query=True)
                    return ops.attn_decode_paged(q, k, v, p["cu_k"], p["block_tables"], p["max_k"], p["scale"],
                                                 cu_seqlens_q=p["cu_q"], out=o)
            o_full = run(p["q"], p["k_cache"], p["v_cache"])
            o_gathered = gather(run(q_l, kc, vc))
            # per-head arithmetic is the same kernel code on the same values, but the launch plan (key
            # split) depends on the number of kv heads per GPU: allow the last-bit effects of a different
            # split, require exact equality where both launches are unsplit
            same = torch.equal(o_full.view(torch.int16), o_gathered.view(torch.int16))
            close = (o_full.float() - o_gathered.float()).abs().max().item() <= 1e-2
            ok = ok and close and (same or max(n_qs) == 1)
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            with open(result_path, "w") as f:
                f.write("ok" if flag.item() == 1.0 else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(N_GPUS < 2, reason="needs at least 2 GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("world", [2, 4, 8])
def test_head_sharded_attention_over_rccl(tmp_path, world):
    if world > N_GPUS:
        pytest.skip(f"{world} ranks need {world} GPUs")
    import torch.multiprocessing as mp
    path = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), path), nprocs=world, join=True)
    assert open(path).read() == "ok"


def _engine_worker(rank, world, port, result_path):
    """Head-sharded engine with HIP-graph decode: tokens equal the unsharded engine's; a prefill step
    LARGER than any before it (which makes the shared all-gather buffers grow) between two runs of decode
    graphs captured earlier must not disturb their replay (the outgrown buffers stay alive)."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", rank)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from nanovllm_amd.engine.llm_engine import LLMEngine
        from nanovllm_amd.engine.sequence import SamplingParams
        from nanovllm_amd.model.qwen3 import ModelConfig, Qwen3ForCausalLM
        from nanovllm_amd.parallel import HeadShard
        cfg = ModelConfig(vocab_size=4096, hidden_size=512, intermediate_size=1024, num_hidden_layers=3,
                          num_attention_heads=16, num_key_value_heads=8, head_dim=128, max_position_embeddings=4096)
        torch.manual_seed(5)
        full = Qwen3ForCausalLM(cfg, layers=None, fused=True)
        for name, prm in full.named_parameters():
            torch.nn.init.normal_(prm, 1.0 if "norm" in name else 0.0, 0.05 if "norm" in name else 0.02)
        full = full.to(torch.bfloat16).to(dev)
        shard = HeadShard(rank, world, cfg.num_attention_heads, cfg.num_key_value_heads)
        part = Qwen3ForCausalLM(cfg, layers=None, fused=True, shard=shard).to(torch.bfloat16).to(dev)
        part.load_full_state_dict(full.state_dict())
        g = torch.Generator().manual_seed(21)
        short = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (16, 40, 100)]
        long_ = [torch.randint(0, cfg.vocab_size, (n,), generator=g).tolist() for n in (700, 650)]
        kw = dict(block_size=256, num_kvcache_blocks=24, max_num_batched_tokens=1500, long_prefill_token_threshold=1024,
                  device=str(dev), eos=-1)
        eng = LLMEngine(part, cfg, use_graphs=True, **kw)
        eager = LLMEngine(part, cfg, **kw)                                   # same sharded model, no graphs
        ref = LLMEngine(full, cfg, **kw)                                     # unsharded
        ok = True
        tok = lambda outs: [o["token_ids"] for o in outs]
        first, _ = eng.generate(short, SamplingParams(0.0, 12))              # captures decode graphs at <= 156 tokens
        gather = part.model.layers[0].self_attn._gather
        buf_small = gather._staging
        second, _ = eng.generate(long_ + short, SamplingParams(0.0, 12))     # a ~1500-token prefill step, then graphs again
        ok = ok and gather._staging is not buf_small and any(t is buf_small for t in gather._retired)
        third, _ = eng.generate(short, SamplingParams(0.0, 12))              # replays the graphs captured first
        # graph replay against the eager run of the SAME sharded model: the same kernels on the same shapes,
        # so the tokens are identical
        e1, _ = eager.generate(short, SamplingParams(0.0, 12))
        e2, _ = eager.generate(long_ + short, SamplingParams(0.0, 12))
        ok = ok and tok(first) == tok(e1) and tok(second) == tok(e2) and tok(third) == tok(e1)
        # against the unsharded model the GEMM shapes differ (column-sharded projections may sum in another
        # order), so a random-init model may flip a token on a numerical tie: first tokens equal, and at
        # least three quarters of all tokens (a wrong head order or a stale buffer gives unrelated tokens)
        w1, _ = ref.generate(short, SamplingParams(0.0, 12))
        w2, _ = ref.generate(long_ + short, SamplingParams(0.0, 12))
        for got, want in ((tok(first), tok(w1)), (tok(second), tok(w2))):
            ok = ok and all(a[0] == b[0] for a, b in zip(got, want))
            same = sum(x == y for a, b in zip(got, want) for x, y in zip(a, b))
            ok = ok and same >= 0.75 * sum(len(a) for a in got)
        ok = ok and len(eng.runner._graphs) >= 1
        # temperature > 0: rank 0's draws are broadcast; every rank must end with the same tokens
        torch.manual_seed(100 + rank)
        sampled, _ = eng.generate(short, SamplingParams(0.8, 10))
        mine = torch.tensor(tok(sampled), device=dev)
        both = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(both, mine)
        ok = ok and all(torch.equal(b, both[0]) for b in both)
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if rank == 0:
            with open(result_path, "w") as f:
                f.write("ok" if flag.item() == 1.0 else "mismatch")
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(N_GPUS < 2, reason="needs at least 2 GPUs (RCCL refuses two ranks on one device)")
def test_head_sharded_engine_with_decode_graphs_over_rccl(tmp_path):
    import torch.multiprocessing as mp
    path = str(tmp_path / "result.txt")
    mp.spawn(_engine_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    assert open(path).read() == "ok"
