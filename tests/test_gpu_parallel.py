"""GPU, needs >= 2 GPUs (skipped on a one-GPU box): head-sharded decode and prefill attention on
BASELINE config 4 shapes (Qwen3-32B: 64 query / 8 kv heads) with one rank per GPU over RCCL; the
all-gathered head outputs must be bit-identical to the single-GPU launch (sharding by kv head leaves
the per-head arithmetic unchanged)."""
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
