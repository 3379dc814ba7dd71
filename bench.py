#!/usr/bin/env python3
"""Benchmark of the MI355X attention hot path on BASELINE.json's headline workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (N=1): Qwen3-0.6B shapes (L=28, H_q=16, H_kv=8, D=128, hidden 1024), decode batch 256,
context lengths ragged U[513,768] (the 512-in/256-out run's range), KV block 256, bf16,
synthetic random data, shuffled physical blocks.  One STEP = one decode step of the hot path
over all 28 layers: per layer two residual-add RMSNorms at hidden size and ONE launch that does the
q/k-norm, the rotary embedding, the KV-cache scatter and the paged decode attention (`--two-launch`: the
scatter launch and the attention launch it replaces).  Inputs are resident in HBM before the timed region.

`value` = algorithmic HBM bytes of the whole step / wall time of the step (GB/s);
`step_pct_of_hbm_peak` is that figure per GPU against 8 TB/s.
`roofline` is for the dominant kernel (paged decode attention) alone: algorithmic bytes per launch
(4096 B per cached token + 8 KiB of Q/O + table bytes, summed over the batch) divided by the
average launch duration measured with HIP events on the launch stream inside the timed steps
(event pairs around the attention launches of every 7th layer, the sampled layers moving from step to
step: a pair around every launch slows the step by 6 %; `attn_kernel_pct_of_hbm_peak` repeats its fraction).
`prefill` (rank 0): the bf16 MFMA prefill kernel on BASELINE config 3 shapes against the 2.5 PFLOP/s peak.
`decode_by_config` (rank 0): the decode launch at the per-GPU shapes of configs 4 and 5.
`cpu_baseline` times the CPU oracle (torch, fp32) of the same attention, and of the scatter / norm /
rotary / prefill kernels, on bounded samples on the host cores (rank 0, N=1 only).

N>1 (launched by torch.distributed.run, one rank per GPU): every rank runs an independent
replica of the same batch (data-parallel replicas, no data-path collective; weak scaling);
the time is the max over ranks between two barriers.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E datasheet peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx-lo", type=int, default=513)
    ap.add_argument("--ctx-hi", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--attn-only", action="store_true", help="step = decode attention launches only")
    ap.add_argument("--two-launch", action="store_true",
                    help="q/k-norm + rotary + scatter as a launch of its own in front of the attention launch")
    ap.add_argument("--tp-layers", type=int, default=8, help="layers in the head-sharded sub-benchmark")
    ap.add_argument("--no-tp", action="store_true", help="skip the head-sharded sub-benchmark")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end generation run (N=1 only)")
    ap.add_argument("--no-prefill", action="store_true", help="skip the prefill-attention records (rank 0)")
    ap.add_argument("--no-configs", action="store_true", help="skip the decode records at config 4 / 5 shapes (rank 0)")
    return ap.parse_args()


class Workload:
    """Qwen3-0.6B-shaped decode step; everything lives on `dev`."""

    H_Q, H_KV, D, HIDDEN, BS = 16, 8, 128, 1024, 256
    EPS, MAX_POS, ROPE_BASE = 1e-6, 40960, 1e6

    def __init__(self, dev, layers, batch, lo, hi, seed=0, long_tail=(), sort_rows=""):
        """`long_tail`: (count, length) pairs that replace the first sequences' lengths (tuning aid: skewed batches)."""
        from nanovllm_amd.layers.rope import build_cos_sin_cache
        self.dev, self.L, self.B = dev, layers, batch
        g = torch.Generator().manual_seed(seed)
        self.ctx = torch.randint(lo, hi + 1, (batch,), generator=g).tolist()
        i = 0
        for count, length in long_tail:
            self.ctx[i:i + count] = [length] * count
            i += count
        if long_tail:                    # the long ones anywhere in the batch
            order = torch.randperm(batch, generator=g).tolist()
            self.ctx = [self.ctx[j] for j in order]
        if sort_rows.startswith("snake"):        # tuning aid: longest first, every other run of N rows reversed ("snake32")
            period = int(sort_rows[5:])
            self.ctx.sort(reverse=True)
            for r in range(1, -(-batch // period), 2):
                self.ctx[r * period:(r + 1) * period] = self.ctx[r * period:(r + 1) * period][::-1]
        elif sort_rows:                  # tuning aid: rows by length ("desc": longest first)
            self.ctx.sort(reverse=sort_rows == "desc")
        blocks = [math.ceil(c / self.BS) for c in self.ctx]
        self.NB = sum(blocks)
        perm = torch.randperm(self.NB, generator=g).tolist()
        width = max(blocks)
        tables, nxt = [], 0
        for nb in blocks:
            tables.append(perm[nxt:nxt + nb] + [-1] * (width - nb))
            nxt += nb
        self.block_tables = torch.tensor(tables, dtype=torch.int32, device=dev)
        self.cu_k = torch.tensor([0] + torch.tensor(self.ctx).cumsum(0).tolist(), dtype=torch.int32, device=dev)
        self.cu_q = torch.arange(batch + 1, dtype=torch.int32, device=dev)
        self.max_k = max(self.ctx)
        # the new token of sequence b sits at position ctx-1
        self.positions = torch.tensor([c - 1 for c in self.ctx], dtype=torch.int64, device=dev)
        self.slots = torch.tensor([tables[b][(c - 1) // self.BS] * self.BS + (c - 1) % self.BS
                                   for b, c in enumerate(self.ctx)], dtype=torch.int32, device=dev)
        # [2, L, NB, bs, H_kv, D] as the reference's runner allocates it, filled with N(0,1)
        self.kv_cache = torch.empty(2, layers, self.NB, self.BS, self.H_KV, self.D, dtype=torch.bfloat16, device=dev)
        for l in range(layers):
            self.kv_cache[0, l].normal_()
            self.kv_cache[1, l].normal_()
        gen = torch.Generator(device=dev).manual_seed(seed)
        qkv_w = (self.H_Q + 2 * self.H_KV) * self.D
        self.qkv = torch.randn(batch, qkv_w, generator=gen, device=dev).bfloat16()
        self.q_in = self.qkv[:, :self.H_Q * self.D].view(batch, self.H_Q, self.D)
        self.k_in = self.qkv[:, self.H_Q * self.D:(self.H_Q + self.H_KV) * self.D].view(batch, self.H_KV, self.D)
        self.v_in = self.qkv[:, (self.H_Q + self.H_KV) * self.D:].view(batch, self.H_KV, self.D)
        self.q_w = (1 + 0.1 * torch.randn(self.D, generator=gen, device=dev)).bfloat16()
        self.k_w = (1 + 0.1 * torch.randn(self.D, generator=gen, device=dev)).bfloat16()
        self.h_w = (1 + 0.1 * torch.randn(self.HIDDEN, generator=gen, device=dev)).bfloat16()
        self.hidden = torch.randn(batch, self.HIDDEN, generator=gen, device=dev).bfloat16()
        self.resid = torch.randn(batch, self.HIDDEN, generator=gen, device=dev).bfloat16()
        self.cos_sin = build_cos_sin_cache(self.D, self.MAX_POS, self.ROPE_BASE).to(dev).view(self.MAX_POS, self.D)
        self.scale = self.D ** -0.5
        # preallocated outputs (the timed region allocates nothing)
        self.q_rot = torch.randn(batch, self.H_Q, self.D, generator=gen, device=dev).bfloat16()
        self.o = torch.empty(batch, self.H_Q, self.D, dtype=torch.bfloat16, device=dev)
        self.y = torch.empty_like(self.hidden)
        self.new_res = torch.empty_like(self.hidden)

    # ---- algorithmic bytes -------------------------------------------------------------
    def attn_bytes_per_launch(self):
        return sum(2 * c * self.H_KV * self.D * 2 + 2 * self.H_Q * self.D * 2 + 4 * math.ceil(c / self.BS) + 8
                   for c in self.ctx)

    def step_bytes(self, attn_only=False, one_launch=True):
        a = self.attn_bytes_per_launch()
        if attn_only:
            return a * self.L
        T = self.B
        heads = self.H_Q + 2 * self.H_KV
        if one_launch:
            # q/k-norm + rotary + scatter inside the attention launch: the raw k, v rows in, the cache rows out,
            # the cos/sin row, position and slot (the raw q row is the attention's own Q read)
            fused = 2 * T * 2 * self.H_KV * self.D * 2 + T * self.D * 4 + 12 * T
        else:
            fused = 2 * T * heads * self.D * 2 + T * self.D * 4 + 12 * T    # in + out, cos/sin row, pos + slot
        norm = T * self.HIDDEN * 2 * 4 + self.HIDDEN * 2                    # x, res in; y, new_res out; weight
        return (a + fused + 2 * norm) * self.L

    # ---- raw C-ABI launch plans (argument tuples prebuilt; nothing allocated per call) ---
    def build_plan(self, attn_only=False, one_launch=True):
        """Per layer: add+RMSNorm, then q/k-norm + rotary + KV scatter + paged decode attention as ONE launch
        (`one_launch`, what the engine's decode step runs) or as the two launches it replaces, then add+RMSNorm."""
        from nanovllm_amd import _lib
        dll = _lib.lib()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        ws_bytes = dll.nv_attn_decode_workspace_bytes(self.B, self.H_Q, self.H_KV, self.D, self.max_k)
        self.ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=self.dev)
        plan = []          # (fn, args, is_attention)
        for l in range(self.L):
            kc, vc = self.kv_cache[0, l], self.kv_cache[1, l]
            norm_args = (self.hidden.data_ptr(), self.resid.data_ptr(), self.h_w.data_ptr(), self.y.data_ptr(),
                         self.new_res.data_ptr(), self.B, self.HIDDEN, self.EPS, stream)
            if not attn_only:
                plan.append((dll.nv_add_rmsnorm, norm_args, False))
            if one_launch and not attn_only:
                plan.append((dll.nv_attn_decode_paged_fused,
                             (self.q_in.data_ptr(), self.k_in.data_ptr(), self.v_in.data_ptr(), self.q_w.data_ptr(),
                              self.k_w.data_ptr(), self.cos_sin.data_ptr(), self.positions.data_ptr(),
                              self.slots.data_ptr(), kc.data_ptr(), vc.data_ptr(), self.o.data_ptr(),
                              self.cu_k.data_ptr(), self.block_tables.data_ptr(), self.B, self.block_tables.shape[1],
                              self.max_k, self.H_Q, self.H_KV, self.D, self.NB, self.BS, self.MAX_POS, self.EPS,
                              self.scale, self.q_in.stride(0), self.k_in.stride(0), self.v_in.stride(0),
                              self.o.stride(0), self.ws.data_ptr(), self.ws.numel(), stream), True))
            else:
                if not attn_only:
                    plan.append((dll.nv_qknorm_rope_store,
                                 (self.q_in.data_ptr(), self.k_in.data_ptr(), self.v_in.data_ptr(), self.q_w.data_ptr(),
                                  self.k_w.data_ptr(), self.cos_sin.data_ptr(), self.positions.data_ptr(),
                                  self.slots.data_ptr(), kc.data_ptr(), vc.data_ptr(), self.q_rot.data_ptr(), None,
                                  self.B, self.H_Q, self.H_KV, self.D, self.MAX_POS, self.EPS, self.NB, self.BS,
                                  self.q_in.stride(0), self.k_in.stride(0), self.v_in.stride(0), stream), False))
                plan.append((dll.nv_attn_decode_paged,
                             (self.q_rot.data_ptr(), kc.data_ptr(), vc.data_ptr(), self.o.data_ptr(),
                              self.cu_q.data_ptr(), self.cu_k.data_ptr(), self.block_tables.data_ptr(), self.B,
                              self.block_tables.shape[1], self.max_k, self.H_Q, self.H_KV, self.D, self.NB, self.BS,
                              self.scale, self.q_rot.stride(0), self.o.stride(0), self.ws.data_ptr(), self.ws.numel(),
                              stream), True))
            if not attn_only:
                plan.append((dll.nv_add_rmsnorm, norm_args, False))
        return plan


def run_step(plan, events=None, phase=0, every=7):
    """One step.  With `events`, HIP event pairs go around the attention launches of every `every`-th layer,
    starting at layer `phase % every` (the caller moves `phase` from step to step, so every layer is sampled):
    a pair around EVERY launch costs the step 6 % (two marker packets per launch: 3.41 against 3.21 ms,
    tools/step_modes.py), which would be charged to `value`."""
    layer = 0
    for fn, args, is_attn in plan:
        if is_attn:
            layer += 1
        if events is not None and is_attn and (layer - 1 + phase) % every == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            rc = fn(*args)
            b.record()
            events.append((a, b))
        else:
            rc = fn(*args)
        if rc != 0:
            from nanovllm_amd import _lib
            _lib.check(rc, fn.__name__)


def cpu_baseline(w: Workload, kc: torch.Tensor, vc: torch.Tensor, seconds: float):
    """The CPU oracle on one layer's cache (host copies kc, vc), sequence by sequence, for about `seconds`."""
    from oracle.ref_attention import paged_attention_ref
    threads = torch.get_num_threads()
    q = torch.randn(w.B, w.H_Q, w.D).bfloat16()
    bt, ctx = w.block_tables.cpu(), w.ctx
    done_bytes, done_seqs = 0, 0
    t0 = time.perf_counter()
    b = 0
    while time.perf_counter() - t0 < seconds:
        i = b % w.B
        paged_attention_ref(q[i:i + 1], kc, vc, torch.tensor([0, 1], dtype=torch.int32),
                            torch.tensor([0, ctx[i]], dtype=torch.int32), bt[i:i + 1], w.scale)
        done_bytes += 2 * ctx[i] * w.H_KV * w.D * 2 + 2 * w.H_Q * w.D * 2 + 4 * math.ceil(ctx[i] / w.BS) + 8
        done_seqs += 1
        b += 1
    dt = time.perf_counter() - t0
    return {"value": round(done_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
            "sample": f"oracle decode attention, 1 layer, {done_seqs} sequences of the same batch "
                      f"(ctx {min(ctx)}..{max(ctx)}), {dt:.1f} s, torch fp32 on {threads} threads"}


def reduce_scalar(dist, dev, value: float, op: str) -> float:
    """MAX / SUM of a python float over the ranks (CUDA tensor for RCCL, host tensor for gloo)."""
    if dist is None:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return t.item()


MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak


def _paged_layout(n_ks, bs, seed, dev):
    """Shuffled physical blocks and -1 padded tables for sequences of n_ks keys."""
    blocks = [math.ceil(n / bs) for n in n_ks]
    NB = sum(blocks)
    g = torch.Generator().manual_seed(seed)
    perm = torch.randperm(NB, generator=g).tolist()
    width = max(blocks)
    tables, nxt = [], 0
    for nb in blocks:
        tables.append(perm[nxt:nxt + nb] + [-1] * (width - nb))
        nxt += nb
    bt = torch.tensor(tables, dtype=torch.int32, device=dev)
    cu_k = torch.tensor([0] + torch.tensor(n_ks).cumsum(0).tolist(), dtype=torch.int32, device=dev)
    return NB, width, bt, cu_k


def _time_launches(fn, n_variants, iters, dev):
    """Average milliseconds of fn(i % n_variants) over `iters` back-to-back launches (HIP events on the
    launch stream, two untimed warm-up launches per variant)."""
    for i in range(2 * n_variants):
        fn(i % n_variants)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(iters):
        fn(i % n_variants)
    b.record()
    torch.cuda.synchronize(dev)
    return a.elapsed_time(b) / iters


def mfma_ceiling_record(dev, launches=5):
    """bf16 MFMA rate this box sustains on v_mfma_f32_32x32x16_bf16 alone (tools/micro/mfma_ceiling.hip, one
    wave per SIMD, register operands): the clock a chip holds under matrix load differs from box to box, so the
    prefill records are priced against this measured ceiling as well as against the nominal peak.  None when
    the probe library has not been built."""
    import ctypes
    path = os.path.join(ROOT, "tools", "micro", "libmfma_ceiling.so")
    if not os.path.exists(path):
        return None
    dll = ctypes.CDLL(path)
    dll.mfma_ceiling_flops.restype = ctypes.c_double
    dll.mfma_ceiling_flops.argtypes = [ctypes.c_int, ctypes.c_int]
    dll.mfma_ceiling_launch.restype = ctypes.c_int
    dll.mfma_ceiling_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    blocks, iters = 1024, 4000
    sink = torch.zeros(blocks * 256, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream

    rates = {}
    for name, rnd in (("random_operands", 1), ("random_operands_16x16x32", 2), ("constant_operands", 0)):
        def launch(_):
            rc = dll.mfma_ceiling_launch(sink.data_ptr(), blocks, iters, rnd, stream)
            if rc != 0:
                raise RuntimeError(f"mfma_ceiling_launch: hip error {rc}")
        ms = _time_launches(launch, 1, launches, dev)
        rates[name] = round(dll.mfma_ceiling_flops(blocks, iters) / (ms * 1e-3) / 1e12, 1)
    # attention data toggles the datapath like the random operands do: that rate is the ceiling used
    return {"achieved_tflops": rates["random_operands"], "achieved_tflops_constant_operands": rates["constant_operands"],
            "achieved_tflops_16x16x32_random_operands": rates["random_operands_16x16x32"],
            "nominal_peak": MFMA_PEAK_TFLOPS,
            "kernel": "tools/micro/mfma_ceiling.hip: back-to-back independent v_mfma_f32_32x32x16_bf16 on register "
                      "operands, 4 waves per workgroup, 1024 workgroups, no memory traffic"}


def prefill_records(dev, iters=10, ceiling_tflops=None):
    """Prefill / chunked-prefill attention (bf16 MFMA kernel) on BASELINE config 3 shapes (Qwen3-8B attention:
    32 query / 8 kv heads x 128, block 256).  flops = 4 * D * H_q * visible (query, key) pairs."""
    from nanovllm_amd import _lib
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    bs, D, H_q, H_kv = 256, 128, 32, 8
    cases = [
        ("one 8192-token prompt prefilled as 16 chunks of 512 (16 launches, keys 512..8192)",
         [([512], [512 * (c + 1)]) for c in range(16)]),
        ("last chunk of that prompt alone: 512 queries over 8192 cached keys", [([512], [8192])]),
        ("7 sequences in one step, a 512-token chunk each, cached keys 512..8192",
         [([512] * 7, [512, 1024, 2048, 4096, 8192, 6144, 3072])]),
        ("the same 7 sequences in the row order the runner gives a step (ModelRunner.step_order: longest first)",
         [([512] * 7, [8192, 6144, 4096, 3072, 2048, 1024, 512])]),
        ("4 sequences x 8192 tokens, whole-prompt prefill", [([8192] * 4, [8192] * 4)]),
        # per-GPU shapes of config 4 (Qwen3-32B, 64 / 8 heads over 8 and 4 GPUs): one sequence has 1 or 2
        # (sequence, kv head) pairs, whose workgroups are spread over all XCDs
        ("one 8192-token prompt, whole-prompt prefill", [([8192], [8192])], (8, 1)),
        ("one 8192-token prompt, whole-prompt prefill", [([8192], [8192])], (16, 2)),
        ("last 512-token chunk of an 8192-token prompt", [([512], [8192])], (8, 1)),
    ]
    out = []
    for case in cases:
        label, launches = case[0], case[1]
        H_q, H_kv = case[2] if len(case) > 2 else (32, 8)
        n_ks_all = launches[-1][1] if len(launches) > 1 else launches[0][1]
        NB, width, bt, _ = _paged_layout(n_ks_all, bs, 3, dev)
        kc = torch.randn(NB, bs, H_kv, D, device=dev).bfloat16()
        vc = torch.randn(NB, bs, H_kv, D, device=dev).bfloat16()
        plans, flops = [], 0
        for n_qs, n_ks in launches:
            B, T = len(n_qs), sum(n_qs)
            q = torch.randn(T, H_q, D, device=dev).bfloat16()
            o = torch.empty_like(q)
            cu_q = torch.tensor([0] + torch.tensor(n_qs).cumsum(0).tolist(), dtype=torch.int32, device=dev)
            cu_k = torch.tensor([0] + torch.tensor(n_ks).cumsum(0).tolist(), dtype=torch.int32, device=dev)
            ws = torch.empty(max(16, dll.nv_attn_prefill_workspace_bytes(B, T, max(n_qs), max(n_ks), H_q, H_kv, D)),
                             dtype=torch.uint8, device=dev)
            plans.append((q, o, cu_q, cu_k, ws, (q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), cu_q.data_ptr(),
                                                cu_k.data_ptr(), bt.data_ptr(), B, width, max(n_qs), max(n_ks), H_q, H_kv, D,
                                                NB, bs, D ** -0.5, q.stride(0), o.stride(0), 0, T, ws.data_ptr(), ws.numel(),
                                                stream)))
            flops += sum(4 * D * H_q * (nq * (nk - nq) + nq * (nq + 1) // 2) for nq, nk in zip(n_qs, n_ks))

        def run_all(_):
            for pl in plans:
                rc = dll.nv_attn_prefill_varlen(*pl[-1])
                if rc != 0:
                    _lib.check(rc, "nv_attn_prefill_varlen")
        ms = _time_launches(run_all, 1, iters, dev)
        tf = flops / (ms * 1e-3) / 1e12
        family = "Qwen3-8B attention shapes" if (H_q, H_kv) == (32, 8) else "Qwen3-32B head-shard shapes"
        out.append({"workload": f"{family} (H {H_q}/{H_kv}, D {D}, block {bs}): {label}",
                    "launches": len(plans), "flops": flops, "ms": round(ms, 4), "achieved_tflops": round(tf, 1),
                    "peak": MFMA_PEAK_TFLOPS, "frac": round(tf / MFMA_PEAK_TFLOPS, 4)})
        if ceiling_tflops:
            out[-1]["frac_of_measured_mfma_ceiling"] = round(tf / ceiling_tflops, 4)
        del plans, kc, vc
        torch.cuda.empty_cache()
    return out


def decode_config_records(dev, iters=40):
    """Decode attention launch at the per-GPU shapes of BASELINE config 4 (Qwen3-32B, 64 query / 8 kv heads,
    batch 256, ctx 2048, kv heads 8/4/2/1 per GPU = 1/2/4/8-way head sharding) and config 5 (Llama-3.1-8B,
    32/8 heads, ctx 32768, 1/4/8 sequences per replica).  Several layers' caches are cycled so that every
    launch streams from HBM (working set > 512 MB)."""
    from nanovllm_amd import _lib
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    bs, D = 256, 128
    shapes = [("config 4", 256, 2048, 8 * hkv, hkv) for hkv in (8, 4, 2, 1)] + \
             [("config 5", b, 32768, 32, 8) for b in (8, 4, 1)]
    out = []
    for cfg_name, B, ctx, H_q, H_kv in shapes:
        n_ks = [ctx] * B
        NB, width, bt, cu_k = _paged_layout(n_ks, bs, 5, dev)
        layer_bytes = 2 * NB * bs * H_kv * D * 2
        L = max(1, min(8, -(-(768 << 20) // layer_bytes)))
        caches = [(torch.randn(NB, bs, H_kv, D, device=dev).bfloat16(), torch.randn(NB, bs, H_kv, D, device=dev).bfloat16())
                  for _ in range(L)]
        q = torch.randn(B, H_q, D, device=dev).bfloat16()
        o = torch.empty_like(q)
        cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev)
        ws = torch.empty(max(16, dll.nv_attn_decode_workspace_bytes(B, H_q, H_kv, D, ctx)), dtype=torch.uint8, device=dev)

        def launch(l):
            rc = dll.nv_attn_decode_paged(q.data_ptr(), caches[l][0].data_ptr(), caches[l][1].data_ptr(), o.data_ptr(),
                                          cu_q.data_ptr(), cu_k.data_ptr(), bt.data_ptr(), B, width, ctx, H_q, H_kv, D, NB,
                                          bs, D ** -0.5, q.stride(0), o.stride(0), ws.data_ptr(), ws.numel(), stream)
            if rc != 0:
                _lib.check(rc, "nv_attn_decode_paged")
        ms = _time_launches(launch, L, iters, dev)
        nbytes = B * (2 * ctx * H_kv * D * 2 + 2 * H_q * D * 2 + 4 * math.ceil(ctx / bs) + 8)
        gbps = nbytes / (ms * 1e-3) / 1e9
        out.append({"workload": f"{cfg_name}: batch {B}, ctx {ctx}, H {H_q}/{H_kv} per GPU, block {bs}",
                    "bytes": nbytes, "ms": round(ms, 5), "achieved_GBps": round(gbps, 1), "peak": HBM_PEAK_GBPS,
                    "frac": round(gbps / HBM_PEAK_GBPS, 4)})

        if B * H_kv <= 8:
            # few (sequence, kv head) pairs: the key ranges combined inside the launch, on a workspace whose arrival
            # counters start at zero and that nothing else writes (what nanovllm_amd.ops gives such launches)
            wz = torch.zeros(max(1 << 16, ws.numel()), dtype=torch.uint8, device=dev)

            def launch_combined(l):
                rc = dll.nv_attn_decode_paged_chunked(q.data_ptr(), caches[l][0].data_ptr(), caches[l][1].data_ptr(),
                                                      o.data_ptr(), cu_q.data_ptr(), cu_k.data_ptr(), bt.data_ptr(), B,
                                                      width, ctx, H_q, H_kv, D, NB, bs, D ** -0.5, q.stride(0),
                                                      o.stride(0), wz.data_ptr(), wz.numel(), 0, 1, stream)
                if rc != 0:
                    _lib.check(rc, "nv_attn_decode_paged_chunked")
            ms_c = _time_launches(launch_combined, L, iters, dev)
            out[-1]["ms_in_launch_combine"] = round(ms_c, 5)
            out[-1]["frac_in_launch_combine"] = round(nbytes / (ms_c * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)

        # the decode step's whole attention block at this shape: the scatter launch (q/k-norm + rotary + KV store)
        # followed by the attention launch, against the one launch that does both (raw projections in)
        q_raw = torch.randn(B, H_q, D, device=dev).bfloat16()
        k_raw = torch.randn(B, H_kv, D, device=dev).bfloat16()
        v_raw = torch.randn(B, H_kv, D, device=dev).bfloat16()
        q_w = (1 + 0.1 * torch.randn(D, device=dev)).bfloat16()
        k_w = (1 + 0.1 * torch.randn(D, device=dev)).bfloat16()
        max_pos = ctx
        cos_sin = torch.randn(max_pos, D, device=dev, dtype=torch.float32)
        positions = torch.full((B,), ctx - 1, dtype=torch.int64, device=dev)
        slots = (bt[:, (ctx - 1) // bs].long() * bs + (ctx - 1) % bs).to(torch.int32)

        def two_launches(l):
            rc = dll.nv_qknorm_rope_store(q_raw.data_ptr(), k_raw.data_ptr(), v_raw.data_ptr(), q_w.data_ptr(),
                                          k_w.data_ptr(), cos_sin.data_ptr(), positions.data_ptr(), slots.data_ptr(),
                                          caches[l][0].data_ptr(), caches[l][1].data_ptr(), q.data_ptr(), None, B, H_q,
                                          H_kv, D, max_pos, 1e-6, NB, bs, q_raw.stride(0), k_raw.stride(0),
                                          v_raw.stride(0), stream)
            if rc != 0:
                _lib.check(rc, "nv_qknorm_rope_store")
            launch(l)

        def one_launch(l):
            rc = dll.nv_attn_decode_paged_fused(q_raw.data_ptr(), k_raw.data_ptr(), v_raw.data_ptr(), q_w.data_ptr(),
                                                k_w.data_ptr(), cos_sin.data_ptr(), positions.data_ptr(),
                                                slots.data_ptr(), caches[l][0].data_ptr(), caches[l][1].data_ptr(),
                                                o.data_ptr(), cu_k.data_ptr(), bt.data_ptr(), B, width, ctx, H_q, H_kv,
                                                D, NB, bs, max_pos, 1e-6, D ** -0.5, q_raw.stride(0), k_raw.stride(0),
                                                v_raw.stride(0), o.stride(0), ws.data_ptr(), ws.numel(), stream)
            if rc != 0:
                _lib.check(rc, "nv_attn_decode_paged_fused")
        try:
            out[-1]["ms_scatter_then_attention"] = round(_time_launches(two_launches, L, iters, dev), 5)
            out[-1]["ms_one_launch"] = round(_time_launches(one_launch, L, iters, dev), 5)
        except Exception as e:              # the attention record above stands on its own
            out[-1]["one_launch_error"] = f"{type(e).__name__}: {e}"
        del caches
        torch.cuda.empty_cache()
    return out


def decode_skew_records(dev, iters=20):
    """The decode launch on skewed batches of the headline shape (256 sequences of 513 .. 767 keys plus a few long
    ones): planned from max_seqlen_k alone (what the reference's call passes) against planned with the chunk
    nv_attn_decode_suggest_chunk derives from the sum of the lengths (what a runner that holds them can pass)."""
    from nanovllm_amd import _lib
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for n_short, label, long_tail in ((256, "one of 8192 keys", [(1, 8192)]), (256, "four of 4096 keys", [(4, 4096)]),
                                      (256, "one of 32768 keys", [(1, 32768)]), (63, "one of 32768 keys", [(1, 32768)])):
        w = Workload(dev, 4, n_short + sum(c for c, _ in long_tail), 513, 767, long_tail=long_tail)
        chunk = dll.nv_attn_decode_suggest_chunk(w.B, w.H_Q, w.H_KV, w.max_k, sum(w.ctx))
        ws = torch.empty(max(16, dll.nv_attn_decode_workspace_bytes_chunked(w.B, w.H_Q, w.H_KV, w.D, w.max_k, chunk),
                             dll.nv_attn_decode_workspace_bytes(w.B, w.H_Q, w.H_KV, w.D, w.max_k)),
                         dtype=torch.uint8, device=dev)

        def launch(l, ck):
            rc = dll.nv_attn_decode_paged_chunked(w.q_rot.data_ptr(), w.kv_cache[0, l].data_ptr(),
                                                  w.kv_cache[1, l].data_ptr(), w.o.data_ptr(), w.cu_q.data_ptr(),
                                                  w.cu_k.data_ptr(), w.block_tables.data_ptr(), w.B,
                                                  w.block_tables.shape[1], w.max_k, w.H_Q, w.H_KV, w.D, w.NB, w.BS,
                                                  w.scale, w.q_rot.stride(0), w.o.stride(0), ws.data_ptr(), ws.numel(),
                                                  ck, 0, stream)
            if rc != 0:
                _lib.check(rc, "nv_attn_decode_paged_chunked")
        ms_plain = _time_launches(lambda l: launch(l, 0), w.L, iters, dev)
        ms_hint = _time_launches(lambda l: launch(l, chunk), w.L, iters, dev)
        nbytes = w.attn_bytes_per_launch()
        out.append({"workload": f"{n_short} sequences of 513..767 keys and {label}, H {w.H_Q}/{w.H_KV}, block {w.BS}",
                    "bytes": nbytes, "ms_planned_from_max": round(ms_plain, 5), "suggested_chunk": chunk,
                    "ms_with_suggested_chunk": round(ms_hint, 5),
                    "GBps_with_suggested_chunk": round(nbytes / (ms_hint * 1e-3) / 1e9, 1)})
        del w, ws
        torch.cuda.empty_cache()
    return out


def decode_row_order_records(dev, iters=10):
    """The decode launch on batches with more work items than the chip holds at once and lengths spread over
    100 .. 2000 keys: rows in random order against longest first (the order ModelRunner.decode_order gives the
    launch; the kernel and its plan are the same)."""
    from nanovllm_amd import _lib
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    out = []
    for batch in (128, 512):
        rec = {"workload": f"{batch} sequences of 100..2000 keys, H {Workload.H_Q}/{Workload.H_KV}, block {Workload.BS}"}
        for key, order in (("ms_random_order", ""), ("ms_longest_first", "desc")):
            w = Workload(dev, 2, batch, 100, 2000, sort_rows=order)
            ws = torch.empty(max(16, dll.nv_attn_decode_workspace_bytes(w.B, w.H_Q, w.H_KV, w.D, w.max_k)),
                             dtype=torch.uint8, device=dev)

            def launch(l):
                rc = dll.nv_attn_decode_paged(w.q_rot.data_ptr(), w.kv_cache[0, l].data_ptr(), w.kv_cache[1, l].data_ptr(),
                                              w.o.data_ptr(), w.cu_q.data_ptr(), w.cu_k.data_ptr(), w.block_tables.data_ptr(),
                                              w.B, w.block_tables.shape[1], w.max_k, w.H_Q, w.H_KV, w.D, w.NB, w.BS, w.scale,
                                              w.q_rot.stride(0), w.o.stride(0), ws.data_ptr(), ws.numel(), stream)
                if rc != 0:
                    _lib.check(rc, "nv_attn_decode_paged")
            rec[key] = round(_time_launches(launch, w.L, iters, dev), 5)
            rec["bytes"] = w.attn_bytes_per_launch()
            del w, ws
            torch.cuda.empty_cache()
        rec["GBps_longest_first"] = round(rec["bytes"] / (rec["ms_longest_first"] * 1e-3) / 1e9, 1)
        out.append(rec)
    return out


def cpu_baseline_other(seconds_each: float = 3.0):
    """The oracle's remaining kernels on the host cores, bounded samples: KV scatter, residual-add RMSNorm,
    rotary embedding (GB/s of the same algorithmic bytes as the GPU kernels) and prefill attention (TFLOP/s)."""
    from oracle import ref_attention, ref_layers
    threads = torch.get_num_threads()
    out = {}

    def rate(fn, per_call):
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds_each:
            fn()
            n += 1
        return n * per_call / (time.perf_counter() - t0)

    T, H_q, H_kv, D, hidden, bs = 4000, 16, 8, 128, 1024, 256
    g = torch.Generator().manual_seed(0)
    k = torch.randn(T, H_kv, D, generator=g).bfloat16()
    v = torch.randn(T, H_kv, D, generator=g).bfloat16()
    kc = torch.zeros(32, bs, H_kv, D, dtype=torch.bfloat16)
    vc = torch.zeros_like(kc)
    slots = torch.randperm(32 * bs, generator=g)[:T].to(torch.int32)
    out["store_kvcache_GBps"] = round(rate(lambda: ref_attention.store_kvcache_ref(k, v, kc, vc, slots, bs),
                                           T * (2 * H_kv * D * 2) * 2 + 4 * T) / 1e9, 3)
    x = torch.randn(T, hidden, generator=g).bfloat16()
    res = torch.randn(T, hidden, generator=g).bfloat16()
    w = torch.ones(hidden).bfloat16()
    out["add_rmsnorm_GBps"] = round(rate(lambda: ref_layers.rmsnorm_ref(x, w, 1e-6, res), T * hidden * 2 * 4 + hidden * 2) / 1e9, 3)
    table = ref_layers.build_cos_sin_cache(D, 4096, 1e6)
    pos = torch.randint(0, 4096, (T,), generator=g)
    q = torch.randn(T, H_q, D, generator=g).bfloat16()
    out["rope_GBps"] = round(rate(lambda: ref_layers.rope_ref(pos, q, k, table),
                                  T * (H_q + H_kv) * D * 2 * 2 + T * D * 4 + 8 * T) / 1e9, 3)
    n = 512                                                   # config 2 prefill: one 512-token sequence, H 16/8
    qp = torch.randn(n, H_q, D, generator=g).bfloat16()
    kc2 = torch.randn(2, bs, H_kv, D, generator=g).bfloat16()
    vc2 = torch.randn(2, bs, H_kv, D, generator=g).bfloat16()
    cu = torch.tensor([0, n], dtype=torch.int32)
    bt = torch.tensor([[0, 1]], dtype=torch.int32)
    flops = 4 * D * H_q * (n * (n + 1) // 2)
    out["prefill_attention_TFLOPs"] = round(rate(lambda: ref_attention.paged_attention_ref(qp, kc2, vc2, cu, cu, bt, D ** -0.5),
                                                 flops) / 1e12, 4)
    out["sample"] = (f"oracle (torch fp32, {threads} threads), ~{seconds_each:.0f} s each: scatter / add+RMSNorm / rotary of "
                     f"{T} tokens (Qwen3-0.6B shapes), prefill attention of one 512-token sequence")
    return out


def _device_identity(dev):
    pr = torch.cuda.get_device_properties(dev)
    ident = {"name": pr.name, "index": dev.index}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(pr, key):
            ident[key] = str(getattr(pr, key))
    return ident


def _gather_objects(dist, world, obj):
    if dist is None:
        return [obj]
    out = [None] * world
    dist.all_gather_object(out, obj)
    return out


def tp_bench(args, dev, rank, world, dist):
    """Head-sharded decode attention + all-gather of head outputs (BASELINE config 4 shapes:
    Qwen3-32B attention, H_q=64, H_kv=8, D=128, batch 256, ctx 2048, block 256).  Rank r holds kv heads
    [r*8/N, (r+1)*8/N) of every layer's cache; metadata is replicated; per layer: local decode
    attention, then one all-gather [T, H_q/N, D] -> [T, H_q, D] and the strided copy into head order.
    A step is the model's 64 launches, cycling over `tp_layers` distinct layer caches (each larger than
    the last-level cache, so every launch streams from HBM).
    The record verifies itself: one more layer is built identically on every rank from a seeded device
    generator (all heads), every rank attends over its shard of it, the head outputs are gathered, and rank 0
    compares them with ONE launch over all heads on its own GPU; the ranks' device identities are gathered to
    show that they are distinct devices."""
    from nanovllm_amd import _lib
    from nanovllm_amd.parallel import HeadShard, all_gather_heads
    H_Q, H_KV, D, BS, B, CTX, L, MODEL_L = 64, 8, 128, 256, 256, 2048, args.tp_layers, 64
    shard = HeadShard(rank, world, H_Q, H_KV)
    hq, hkv = shard.local_heads, shard.local_kv_heads
    g = torch.Generator().manual_seed(1234)                      # same tables on every rank
    nb_seq = CTX // BS
    NB = B * nb_seq
    perm = torch.randperm(NB, generator=g)
    bt = perm.view(B, nb_seq).to(torch.int32).to(dev)
    cu_k = (torch.arange(B + 1, dtype=torch.int32) * CTX).to(dev)
    cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev)
    cache = torch.empty(2, L, NB, BS, hkv, D, dtype=torch.bfloat16, device=dev)
    for l in range(L):
        cache[0, l].normal_()
        cache[1, l].normal_()
    q = torch.randn(B, hq, D, device=dev).bfloat16()
    o_local = torch.empty_like(q)
    o_full = torch.empty(B, H_Q, D, dtype=torch.bfloat16, device=dev)
    staging = torch.empty(world * B, hq, D, dtype=torch.bfloat16, device=dev)
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(max(dll.nv_attn_decode_workspace_bytes(B, hq, hkv, D, CTX), 16), dtype=torch.uint8, device=dev)
    gloo = dist is not None and dist.get_backend() == "gloo"

    def attend(qt, kc, vc, ot, heads_q, heads_kv, wsp):
        rc = dll.nv_attn_decode_paged(qt.data_ptr(), kc.data_ptr(), vc.data_ptr(), ot.data_ptr(),
                                      cu_q.data_ptr(), cu_k.data_ptr(), bt.data_ptr(), B, nb_seq, CTX, heads_q, heads_kv, D,
                                      NB, BS, D ** -0.5, qt.stride(0), ot.stride(0), wsp.data_ptr(), wsp.numel(), stream)
        if rc != 0:
            _lib.check(rc, "nv_attn_decode_paged")

    def step(mode):
        """mode 0: attention only; 1: + all-gather into the rank-major staging buffer; 2: + the copy into head order."""
        for i in range(MODEL_L):
            l = i % L
            attend(q, cache[0, l], cache[1, l], o_local, hq, hkv, ws)
            if mode and gloo:            # rehearsal only: gloo gathers host tensors
                o_full.copy_(all_gather_heads(o_local.cpu(), shard))
            elif mode == 1:
                dist.all_gather_into_tensor(staging, o_local)
            elif mode == 2:
                all_gather_heads(o_local, shard, out=o_full, staging=staging)

    def timed(mode):
        for _ in range(2):
            step(mode)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(mode)
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        return reduce_scalar(dist, dev, time.perf_counter() - t0, "max") / args.steps

    t_attn = timed(0)
    t_gather = timed(1) if world > 1 else t_attn
    t_full = timed(2) if world > 1 else t_attn
    del cache
    torch.cuda.empty_cache()

    # ---- self-check on one more layer, identical on every rank ----
    gen = torch.Generator(device=dev).manual_seed(4321)
    k_all = torch.randn(NB, BS, H_KV, D, generator=gen, device=dev, dtype=torch.float32).bfloat16()
    v_all = torch.randn(NB, BS, H_KV, D, generator=gen, device=dev, dtype=torch.float32).bfloat16()
    q_all = torch.randn(B, H_Q, D, generator=gen, device=dev, dtype=torch.float32).bfloat16()
    k_my, v_my = shard.slice_kv(k_all).contiguous(), shard.slice_kv(v_all).contiguous()
    attend(shard.slice_q(q_all), k_my, v_my, o_local, hq, hkv, ws)
    if world == 1:
        gathered = o_local
    elif gloo:
        gathered = all_gather_heads(o_local.cpu(), shard).to(dev)
    else:
        gathered = all_gather_heads(o_local, shard, out=o_full, staging=staging)
    verified, max_diff, bit_identical = True, 0.0, True
    if rank == 0:
        want = torch.empty(B, H_Q, D, dtype=torch.bfloat16, device=dev)
        ws_all = torch.empty(max(dll.nv_attn_decode_workspace_bytes(B, H_Q, H_KV, D, CTX), 16), dtype=torch.uint8, device=dev)
        attend(q_all, k_all, v_all, want, H_Q, H_KV, ws_all)
        torch.cuda.synchronize(dev)
        max_diff = (want.float() - gathered.float()).abs().max().item()
        bit_identical = bool(torch.equal(want.view(torch.int16), gathered.view(torch.int16)))
        # the launch plan (key split) follows the kv heads per GPU, so the fp32 summation order may differ
        verified = bool(max_diff <= 1e-2 and torch.isfinite(gathered.float()).all())
    del k_all, v_all, k_my, v_my
    torch.cuda.empty_cache()
    devices = _gather_objects(dist, world, _device_identity(dev))
    ids = [d.get("uuid") or d.get("pci_bus_id") or f"index {d.get('index')}" for d in devices]
    bytes_layer_all = B * (2 * CTX * H_KV * D * 2 + 2 * H_Q * D * 2 + 4 * nb_seq + 8)     # all ranks together
    if dist is not None:
        assert dist.get_world_size() == world == args.gpus, "process group size differs from --gpus"
    rec = {"world_size": world,
           "verified": verified, "verify_max_abs_diff": round(max_diff, 6), "verify_bit_identical": bit_identical,
           "verify": "one layer built identically on every rank (seeded device generator): gathered head outputs of the "
                     "sharded launches against one launch over all 64 / 8 heads on rank 0 (bound 1e-2)",
           "devices": devices, "distinct_devices": len(set(ids)),
           "ms_allgather_per_layer": round(max(t_gather - t_attn, 0.0) / MODEL_L * 1e3, 4),
           "ms_permute_per_layer": round(max(t_full - t_gather, 0.0) / MODEL_L * 1e3, 4),
           "workload": f"Qwen3-32B attention shapes, decode batch {B}, ctx {CTX}, block {BS}, {MODEL_L} launches per step "
                       f"over {L} distinct layer caches, kv heads sharded {H_KV}/{world} per GPU, RCCL all-gather of o per layer",
           "ms_per_layer_attn_only": round(t_attn / MODEL_L * 1e3, 4),
           "ms_per_layer_with_allgather": round(t_full / MODEL_L * 1e3, 4),
           "ms_per_step_64_layers": round(t_full * 1e3, 3),
           "attn_GBps_all_gpus": round(bytes_layer_all * MODEL_L / t_full / 1e9, 1),
           "allgather_bytes_per_layer": B * H_Q * D * 2,
           "tok_per_s_attention_path": round(B / t_full, 1)}
    if rank == 0 and not verified:
        rec["error"] = f"gathered head outputs differ from the all-heads launch by {max_diff}"
    return rec


def dp_config5_bench(args, dev, rank, world, dist):
    """BASELINE config 5 as data-parallel replicas: every rank decodes its own batch of 8 sequences x 32768
    cached tokens at Llama-3.1-8B attention shapes (32 query / 8 kv heads x 128, block 256), no collective on
    the data path; value = bytes of all ranks / max time over ranks.  Two layer caches are cycled (1 GiB
    each), 32 launches (the model's layers) per step."""
    from nanovllm_amd import _lib
    dll = _lib.lib()
    stream = torch.cuda.current_stream(dev).cuda_stream
    B, ctx, H_q, H_kv, D, bs, L, MODEL_L = 8, 32768, 32, 8, 128, 256, 2, 32
    NB, width, bt, cu_k = _paged_layout([ctx] * B, bs, 100 + rank, dev)
    caches = [(torch.randn(NB, bs, H_kv, D, device=dev).bfloat16(), torch.randn(NB, bs, H_kv, D, device=dev).bfloat16())
              for _ in range(L)]
    q = torch.randn(B, H_q, D, device=dev).bfloat16()
    o = torch.empty_like(q)
    cu_q = torch.arange(B + 1, dtype=torch.int32, device=dev)
    ws = torch.empty(max(16, dll.nv_attn_decode_workspace_bytes(B, H_q, H_kv, D, ctx)), dtype=torch.uint8, device=dev)

    def step():
        for i in range(MODEL_L):
            kc, vc = caches[i % L]
            rc = dll.nv_attn_decode_paged(q.data_ptr(), kc.data_ptr(), vc.data_ptr(), o.data_ptr(), cu_q.data_ptr(),
                                          cu_k.data_ptr(), bt.data_ptr(), B, width, ctx, H_q, H_kv, D, NB, bs, D ** -0.5,
                                          q.stride(0), o.stride(0), ws.data_ptr(), ws.numel(), stream)
            if rc != 0:
                _lib.check(rc, "nv_attn_decode_paged")
    for _ in range(2):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    steps = max(2, args.steps // 4)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    t = reduce_scalar(dist, dev, time.perf_counter() - t0, "max") / steps
    finite = bool(torch.isfinite(o.float()).all())
    nbytes = B * (2 * ctx * H_kv * D * 2 + 2 * H_q * D * 2 + 4 * math.ceil(ctx / bs) + 8) * MODEL_L
    total = reduce_scalar(dist, dev, float(nbytes), "sum")
    del caches
    torch.cuda.empty_cache()
    return {"workload": f"config 5: {world} replica(s) x batch {B} x ctx {ctx}, H {H_q}/{H_kv}, block {bs}, "
                        f"{MODEL_L} decode-attention launches per step over {L} layer caches",
            "replicas": world, "ms_per_step": round(t * 1e3, 3), "attn_GBps_all_gpus": round(total / t / 1e9, 1),
            "frac_of_hbm_peak_per_gpu": round(total / t / 1e9 / world / HBM_PEAK_GBPS, 4),
            "tok_per_s_attention_path": round(B * world / t, 1), "outputs_finite": finite}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("for --gpus N > 1 launch with: python -m torch.distributed.run --nnodes=1 "
                             "--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU implementation)")
    # Rehearsal aid for boxes with fewer GPUs than ranks: NV_BENCH_SHARE_GPU=1 puts every rank on
    # cuda:0 and uses gloo (RCCL refuses two ranks on one device).  Never used for reported numbers.
    share = os.environ.get("NV_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            import datetime
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    w = Workload(dev, args.layers, args.batch, args.ctx_lo, args.ctx_hi, seed=rank)
    one_launch = not args.two_launch and not args.attn_only
    plan = w.build_plan(args.attn_only, one_launch)

    def barrier():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        run_step(plan)
    barrier()
    events = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        run_step(plan, events, phase=i, every=min(7, args.layers))
    barrier()
    elapsed = reduce_scalar(dist, dev, time.perf_counter() - t0, "max")
    step_bytes_all = reduce_scalar(dist, dev, float(w.step_bytes(args.attn_only, one_launch)), "sum")

    ms_per_step = elapsed / args.steps * 1e3
    value = step_bytes_all / (elapsed / args.steps) / 1e9
    attn_ms = sum(a.elapsed_time(b) for a, b in events) / len(events)
    achieved = w.attn_bytes_per_launch() / (attn_ms * 1e-3) / 1e9
    # beside the one-launch step: the attention launch on its own (rotated queries in, nothing stored), three
    # untimed-for-`value` passes over the layers with an event pair around every launch
    attn_alone = None
    if one_launch and rank == 0:
        plan_a = w.build_plan(True, False)
        ev_a = []
        run_step(plan_a)
        for _ in range(3):
            run_step(plan_a, ev_a, every=1)
        torch.cuda.synchronize(dev)
        ms_a = sum(a.elapsed_time(b) for a, b in ev_a) / len(ev_a)
        gbps_a = w.attn_bytes_per_launch() / (ms_a * 1e-3) / 1e9
        attn_alone = {"kernel": "attn_decode_kernel (nv_attn_decode_paged: the attention launch without the fused "
                                "q/k-norm, rotary and scatter)", "launch_ms": round(ms_a, 5),
                      "achieved": round(gbps_a, 1), "unit": "GB/s", "frac": round(gbps_a / HBM_PEAK_GBPS, 4)}
        del plan_a

    want_cpu = rank == 0 and world == 1 and not args.no_cpu_baseline
    cpu_kv = (w.kv_cache[0, 0].cpu(), w.kv_cache[1, 0].cpu()) if want_cpu else None
    del plan
    w.kv_cache = None
    torch.cuda.empty_cache()
    # The records below are additions to the line: a failure in one of them is reported in its place and
    # never costs the line itself (an exception raised from the same arguments is raised on every rank alike,
    # so no rank is left waiting in a collective).
    def guarded(fn, *a, **kw):
        try:
            return fn(*a, **kw)
        except Exception as e:      # noqa: BLE001
            return {"error": f"{type(e).__name__}: {e}"}

    tp = None if args.no_tp else guarded(tp_bench, args, dev, rank, world, dist)
    dp5 = None if args.no_tp else guarded(dp_config5_bench, args, dev, rank, world, dist)

    extra_prefill = extra_cfg = extra_skew = extra_order = ceiling = None
    if rank == 0 and not args.no_prefill:
        ceiling = guarded(mfma_ceiling_record, dev)
        ceil_tf = ceiling.get("achieved_tflops") if isinstance(ceiling, dict) else None
        extra_prefill = guarded(prefill_records, dev, ceiling_tflops=ceil_tf)
    if rank == 0 and not args.no_configs:
        extra_cfg = guarded(decode_config_records, dev)
        extra_skew = guarded(decode_skew_records, dev)
        extra_order = guarded(decode_row_order_records, dev)

    if rank == 0:
        # HBM traffic of the dominant kernel comes from separate rocprofv3 --pmc passes (counters cannot be
        # read inside this process); it is reported only while the kernel source and the workload are the
        # ones the passes were taken on.
        traffic, traffic_source = None, None
        tf = os.path.join(ROOT, "profiles", "decode_traffic.json")
        if os.path.exists(tf):
            try:
                rec = json.load(open(tf))
                import hashlib
                src = os.path.join(ROOT, "nano-vllm_amd", "csrc", "attn_decode.hip")
                sha = hashlib.sha256(open(src, "rb").read()).hexdigest()
                default_wl = (args.batch, args.ctx_lo, args.ctx_hi) == (256, 513, 768)
                # (the passes are taken on the one-launch step's kernel)
                if rec.get("kernel_source_sha256") == sha and default_wl and one_launch:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = {"file": "profiles/decode_traffic.json", "commit": rec.get("commit"),
                                      "kernel_source_sha256": sha[:16], "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, "
                                      "separate passes, FETCH_SIZE doubled (gfx950)"}
                else:
                    traffic_source = {"file": "profiles/decode_traffic.json", "stale": True,
                                      "reason": "kernel source, launch form or workload differs from the PMC passes"}
            except Exception:
                traffic = None
        out = {
            "metric": "paged-decode attn HBM GB/s", "value": round(value, 1), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "value_scope": "whole decode step of the hot path (all kernels of all layers): algorithmic bytes / wall time",
            "config": {"workload": f"Qwen3-0.6B shapes, decode batch {args.batch}, ctx U[{args.ctx_lo},{args.ctx_hi}], "
                                   f"block 256, {args.layers} layers; step = "
                                   + ("decode attention only" if args.attn_only else
                                      ("2x add+RMSNorm and one launch of qk-norm+RoPE+KV store+paged decode attention per layer"
                                       if one_launch else
                                       "2x add+RMSNorm, fused qk-norm+RoPE+KV store, paged decode attention per layer")),
                       "H_q": 16, "H_kv": 8, "head_dim": 128, "batch": args.batch, "layers": args.layers,
                       "parallelism": "single GPU" if world == 1 else f"dp{world} replicas"
                                      + (" (REHEARSAL: ranks share one GPU, gloo)" if share else ""),
                       "world_size_checked": world if dist is None else dist.get_world_size()},
            "step_pct_of_hbm_peak": round(100 * value / world / HBM_PEAK_GBPS, 2),
            "attn_kernel_pct_of_hbm_peak": round(100 * achieved / HBM_PEAK_GBPS, 2),
            "hot_path_tok_per_s": round(args.batch * world / (ms_per_step * 1e-3), 1),
            "roofline": {"bound": "hbm", "kernel": ("attn_decode_kernel<FUSE> (the attention launch, which also normalises, rotates and stores "
                                    "the step's token; priced on the attention's algorithmic bytes alone; rank 0)"
                                    if one_launch else "attn_decode_kernel (decode attention launch only, rank 0)"),
                         "achieved": round(achieved, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic, "traffic_source": traffic_source, "launch_ms": round(attn_ms, 5),
                         "algorithmic_bytes_per_launch": w.attn_bytes_per_launch()},
        }
        if attn_alone is not None:
            out["attention_launch_alone"] = attn_alone
        if extra_prefill is not None:
            out["prefill"] = extra_prefill
            if ceiling is not None:
                out["mfma_ceiling"] = ceiling
        if extra_cfg is not None:
            out["decode_by_config"] = extra_cfg
        if extra_skew is not None:
            out["decode_skewed_batches"] = extra_skew
        if extra_order is not None:
            out["decode_row_order"] = extra_order
        if tp is not None:
            out["head_sharded"] = tp
        if dp5 is not None:
            out["dp_config5"] = dp5
        if want_cpu:
            out["cpu_baseline"] = cpu_baseline(w, cpu_kv[0], cpu_kv[1], args.cpu_seconds)
            out["cpu_baseline"]["other_kernels"] = cpu_baseline_other()
        if world == 1 and not args.no_e2e:
            # BASELINE config 2 end to end: 256 prompts x 512 tokens in, 256 out, argmax, random-init
            # Qwen3-0.6B-shaped weights; GEMMs are torch/hipBLASLt, the attention path is this library.
            from tools.e2e_bench import run as e2e_run
            del w
            torch.cuda.empty_cache()
            e = guarded(e2e_run, device=str(dev))
            if "error" not in e:
                out["output_tok_per_s"] = round(e["output_tok_per_s"], 1)
            out["e2e"] = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()}
        print(json.dumps(out), flush=True)
        out = None
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
