#!/usr/bin/env python3
"""Tuning aid: time the paged decode attention launch of bench.py's workload for several library
variants (nano-vllm_amd/lib/variants/lib_*.so) and key-chunk sizes, interleaved in one process.

    python tools/decode_sweep.py --chunks 0,128,256,512 --rounds 5 [--variants a,b]
"""
import argparse
import glob
import json
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nanovllm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", default="0", help="comma list; 0 = heuristic; negative -N = heuristic with target N waves")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="")
    ap.add_argument("--layers", type=int, default=28)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--ctx-lo", type=int, default=513)
    ap.add_argument("--ctx-hi", type=int, default=768)
    ap.add_argument("--hq", type=int, default=16)
    ap.add_argument("--hkv", type=int, default=8)
    ap.add_argument("--long", default="", help="skewed batch: comma list of COUNT:LENGTH replacing the first sequences")
    ap.add_argument("--chunk-by-call", action="store_true", help="positive chunks go in as nv_attn_decode_paged_chunked's argument instead of the process-wide override")
    ap.add_argument("--hint", action="store_true", help="add a row per library with the chunk nv_attn_decode_suggest_chunk gives for the batch's lengths")
    ap.add_argument("--sort", default="", help="order the batch's sequences by length: desc, asc, or snakeN (longest first, every other run of N rows reversed)")
    ap.add_argument("--no-cu-q", action="store_true", help="pass no cu_seqlens_q (token b is sequence b's)")
    ap.add_argument("--modes", default="auto", help="comma list of auto,valu,mfma0,mfma1 (G >= 4 kernel choice),wgs0,wgs1 (work item = wave / workgroup),single0,single1 (merge launch / last-arriver merge),static (allocation-free entry point),fused (one-launch step)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    bench.Workload.H_Q, bench.Workload.H_KV = args.hq, args.hkv
    long_tail = [tuple(int(x) for x in item.split(":")) for item in args.long.split(",") if item]
    w = bench.Workload(dev, args.layers, args.batch, args.ctx_lo, args.ctx_hi, long_tail=long_tail, sort_rows=args.sort)
    libs = {"default": _lib.LIB_PATH}
    for path in sorted(glob.glob(os.path.join(ROOT, "nano-vllm_amd", "lib", "variants", "lib_*.so"))):
        libs[os.path.basename(path)[4:-3]] = path
    if args.variants:
        libs = {k: v for k, v in libs.items() if k in args.variants.split(",")}
    chunks = [int(c) for c in args.chunks.split(",")]
    if args.hint:
        hint = _lib.lib().nv_attn_decode_suggest_chunk(w.B, w.H_Q, w.H_KV, w.max_k, sum(w.ctx))
        print(f"suggested chunk: {hint}")
        if hint and hint not in chunks:
            chunks.append(hint)
    stream = torch.cuda.current_stream(dev).cuda_stream
    ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
    results = {}
    dlls = {name: _lib.load(path) for name, path in libs.items()}
    for rnd in range(args.rounds + 1):
        for name, dll in dlls.items():
          for mode in args.modes.split(","):
            fields = {}
            if mode in ("wgs0", "wgs1"):
                fields["decode_work_item"] = int(mode[3])
            elif mode == "valu":
                fields["decode_score_path"] = 0
            elif mode.startswith("mfma"):
                fields["decode_mfma_depth"] = int(mode[4:])
            elif mode in ("single0", "single1"):
                fields["decode_single_launch"] = int(mode[6])
            for chunk in chunks:
                f = dict(fields)
                if chunk > 0 and not args.chunk_by_call:
                    f["decode_chunk"] = chunk
                elif chunk < 0:
                    f["decode_target_waves"] = -chunk
                t = _lib.Tuning(**{**_lib.Tuning.DEFAULTS, **f})
                assert dll.nv_set_tuning(t) == 0, dll.nv_last_error()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for l in range(w.L):
                    args_ = (w.q_rot.data_ptr(), w.kv_cache[0, l].data_ptr(), w.kv_cache[1, l].data_ptr(),
                             w.o.data_ptr(), None if args.no_cu_q else w.cu_q.data_ptr(), w.cu_k.data_ptr(), w.block_tables.data_ptr(), w.B,
                             w.block_tables.shape[1], w.max_k, w.H_Q, w.H_KV, w.D, w.NB, w.BS, w.scale,
                             w.q_rot.stride(0), w.o.stride(0))
                    if mode == "static":       # the allocation-free entry point (in-launch combine by default)
                        rc = dll.nv_attn_decode_paged_static(*args_, stream)
                    elif mode == "fused":      # the one-launch step: raw projections in, token stored by the launch
                        fargs = (w.q_in.data_ptr(), w.k_in.data_ptr(), w.v_in.data_ptr(), w.q_w.data_ptr(), w.k_w.data_ptr(),
                                 w.cos_sin.data_ptr(), w.positions.data_ptr(), w.slots.data_ptr(),
                                 w.kv_cache[0, l].data_ptr(), w.kv_cache[1, l].data_ptr(), w.o.data_ptr(), w.cu_k.data_ptr(),
                                 w.block_tables.data_ptr(), w.B, w.block_tables.shape[1], w.max_k, w.H_Q, w.H_KV, w.D, w.NB,
                                 w.BS, w.MAX_POS, w.EPS, w.scale, w.q_in.stride(0), w.k_in.stride(0), w.v_in.stride(0),
                                 w.o.stride(0), ws.data_ptr(), ws.numel())
                        if args.chunk_by_call and chunk > 0:
                            rc = dll.nv_attn_decode_paged_fused_chunked(*fargs, chunk, 0, stream)
                        else:
                            rc = dll.nv_attn_decode_paged_fused(*fargs, stream)
                    elif args.chunk_by_call and chunk > 0:    # the host's chunk as an argument (what a runner passes)
                        rc = dll.nv_attn_decode_paged_chunked(*args_, ws.data_ptr(), ws.numel(), chunk, 0, stream)
                    else:
                        rc = dll.nv_attn_decode_paged(*args_, ws.data_ptr(), ws.numel(), stream)
                    assert rc == 0, dll.nv_last_error()
                b.record()
                torch.cuda.synchronize()
                if rnd > 0:
                    results.setdefault((name + ":" + mode, chunk), []).append(a.elapsed_time(b) / w.L)
    nbytes = w.attn_bytes_per_launch()
    print(f"workload: B={w.B} H_q={w.H_Q} H_kv={w.H_KV} ctx {min(w.ctx)}..{max(w.ctx)} bytes/launch={nbytes}")
    for (name, chunk), ts in sorted(results.items(), key=lambda kv: statistics.median(kv[1])):
        med, best = statistics.median(ts), min(ts)
        print(f"{name:>16s} chunk={chunk:4d}  median {med*1e3:8.2f} us  {nbytes/med/1e6:7.1f} GB/s   "
              f"best {best*1e3:8.2f} us {nbytes/best/1e6:7.1f} GB/s")
    for dll in dlls.values():
        dll.nv_set_tuning(None)

if __name__ == "__main__":
    main()
