#!/usr/bin/env python3
"""Copy the summaries of one tools/gpu_round.sh session (gpurun_out/<session>/) into profiles/ under the round's
names, and rebuild profiles/decode_traffic.json from its two counter passes.

    python tools/collect_profiles.py r03b r03        # session directory, file-name prefix
"""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(pattern):
    hits = sorted(glob.glob(pattern, recursive=True))
    if not hits:
        sys.exit(f"nothing matches {pattern}")
    return hits[0]


def main():
    session, prefix = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", session)
    dst = os.path.join(ROOT, "profiles")

    def put(pattern, name):
        shutil.copyfile(one(os.path.join(src, pattern)), os.path.join(dst, f"{prefix}_{name}"))

    put("bench_line.json", "bench_line.json")
    put("bench_line_under_rocprof.json", "bench_line_under_rocprof.json")
    put("stats/**/*kernel_stats.csv", "bench_kernel_stats.csv")
    put("pmc_fetch/**/*counter_collection.csv", "pmc_fetch_size.csv")
    put("pmc_write/**/*counter_collection.csv", "pmc_write_size.csv")
    put("pmc_fetch_h8_1/**/*counter_collection.csv", "pmc_fetch_size_h8_1_shard.csv")
    put("pmc_write_h8_1/**/*counter_collection.csv", "pmc_write_size_h8_1_shard.csv")
    put("prefill_stats/**/*kernel_stats.csv", "prefill_kernel_stats.csv")
    put("prefill_stats.out", "prefill_under_rocprof.txt")
    put("pmc_mfma/**/*counter_collection.csv", "prefill_pmc_mfma_busy.csv")
    put("chunk_sweep.txt", "chunk_sweep.txt")
    put("prefill_by_case.txt", "prefill_by_case.txt")
    put("dec_1x32k.out", "decode_1x32k.txt")
    put("dec_1x32k/**/*kernel_stats.csv", "decode_1x32k_kernel_stats.csv")
    # the per-dispatch trace of the 16 chunks, reduced to the columns that matter
    keep = ["Dispatch_Id", "Kernel_Name", "Workgroup_Size_X", "Grid_Size_X", "LDS_Block_Size", "VGPR_Count",
            "Accum_VGPR_Count", "Start_Timestamp", "End_Timestamp"]
    with open(one(os.path.join(src, "chunk_trace/**/*kernel_trace.csv")), newline="") as f, \
            open(os.path.join(dst, f"{prefix}_chunk_kernel_trace.csv"), "w", newline="") as g:
        w = csv.writer(g)
        w.writerow(keep + ["Duration_ns"])
        for row in csv.DictReader(f):
            if "prefill" not in row["Kernel_Name"]:
                continue
            w.writerow([row[k] for k in keep] + [int(row["End_Timestamp"]) - int(row["Start_Timestamp"])])
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, text=True).strip()
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_traffic_json.py"),
                           "--fetch", os.path.join(src, "pmc_fetch"), "--write", os.path.join(src, "pmc_write"),
                           "--commit", commit])


if __name__ == "__main__":
    main()
