#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately: they do not fit one
pass on gfx950) of `bench.py` into profiles/decode_traffic.json.  The kernel is the bench step's attention
launch, which also normalises, rotates and stores the step's token (FUSE instantiation of attn_decode_kernel).

    python tools/make_traffic_json.py --fetch DIR_WITH_FETCH_CSV --write DIR_WITH_WRITE_CSV --commit HASH

HBM bytes per launch = 2 * FETCH_SIZE (gfx950 reports half the bytes of a 16-B-per-lane coalesced stream)
+ WRITE_SIZE, both in KiB, averaged over the dispatches of the decode kernel.  The record carries the SHA-256
of csrc/attn_decode.hip so that bench.py can tell when the kernel has changed since the passes were taken."""
import argparse
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_avg(directory, counter, kernel_substr):
    vals = {}
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter or kernel_substr not in row.get("Kernel_Name", ""):
                    continue
                vals.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    if not vals:
        sys.exit(f"no {counter} rows for a kernel containing {kernel_substr!r} under {directory}")
    name = max(vals, key=lambda k: len(vals[k]))
    return name, sum(vals[name]) / len(vals[name]), len(vals[name])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--commit", default="")
    ap.add_argument("--kernel", default="attn_decode_kernel<2, true, 0, false, false, true>",
                    help="substring of the kernel name (template arguments: G, ALIGNED, depth, WGS, TK, FUSE)")
    ap.add_argument("--algorithmic-bytes", type=int, default=667157504,
                    help="attention bytes of one launch of the bench workload")
    ap.add_argument("--fused-extra-bytes", type=int, default=2 * 256 * 2 * 8 * 128 * 2 + 256 * 128 * 4 + 12 * 256,
                    help="what the fused launch moves beside them: raw k, v rows in, cache rows out, cos/sin row, "
                         "position and slot per token")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "decode_traffic.json"))
    a = ap.parse_args()
    name, fetch_kb, n = counter_avg(a.fetch, "FETCH_SIZE", a.kernel)
    _, write_kb, _ = counter_avg(a.write, "WRITE_SIZE", a.kernel)
    sha = hashlib.sha256(open(os.path.join(ROOT, "nano-vllm_amd", "csrc", "attn_decode.hip"), "rb").read()).hexdigest()
    rec = {
        "kernel": name,
        "workload": "bench.py default (Qwen3-0.6B shapes, B=256, ctx U[513,768], block 256)",
        "FETCH_SIZE_KB_avg": round(fetch_kb, 1), "WRITE_SIZE_KB_avg": round(write_kb, 1),
        "correction": "gfx950: FETCH_SIZE reads half of the bytes of a 16-B/lane coalesced stream -> doubled; WRITE_SIZE exact",
        "hbm_bytes_per_launch": int(round((2 * fetch_kb + write_kb) * 1024)),
        "algorithmic_bytes_per_launch": a.algorithmic_bytes,
        "fused_extra_bytes_per_launch": a.fused_extra_bytes,
        "dispatches_averaged": n,
        "kernel_source_sha256": sha,
        "commit": a.commit,
        "commands": ["rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 2 --warmup 1 "
                     "--no-cpu-baseline --no-e2e --no-tp --no-prefill --no-configs",
                     "same with --pmc WRITE_SIZE"],
    }
    json.dump(rec, open(a.out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
