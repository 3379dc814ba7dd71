#!/bin/bash
# One GPU session: parity tests, the bench line, the kernel-trace profile of the same command, PMC passes.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
OUT=gpurun_out/${1:-r03}
BENCH_FLAGS=${2:-}          # e.g. --no-e2e
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python -u bench.py $BENCH_FLAGS > $OUT/bench_line.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
cat $OUT/bench_line.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-tp > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof_stats.err || { tail -30 $OUT/rocprof_stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-tp --no-prefill --no-configs > $OUT/pmc_fetch.out 2>&1 || { tail -30 $OUT/pmc_fetch.out; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-tp --no-prefill --no-configs > $OUT/pmc_write.out 2>&1 || { tail -30 $OUT/pmc_write.out; exit 1; }
# the hand-scheduled prefill kernel alone: kernel trace, then MFMA-busy against wave cycles
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefill_stats -- python3 tools/prefill_bench.py --kernels 1 --cases b4_n8192_h32,b16_n2048_h32 --rounds 2 --iters 3 > $OUT/prefill_stats.out 2>&1 || { tail -30 $OUT/prefill_stats.out; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 tools/prefill_bench.py --kernels 1 --cases b4_n8192_h32 --rounds 1 --iters 2 > $OUT/pmc_mfma.out 2>&1 || { tail -30 $OUT/pmc_mfma.out; exit 1; }
# chunked prefill: the 16 chunks of one 8192-token prompt, launch plan's choice, with a per-dispatch kernel trace
CHUNKS=$(python3 -c "print(','.join(f'chunk_512x{512*(c+1)}' for c in range(16)))")
timeout -k 10 300 python3 tools/prefill_bench.py --kernels auto --rounds 3 --iters 4 --cases $CHUNKS,cfg3_7seq_chunks > $OUT/chunk_sweep.txt 2>/dev/null || { tail -5 $OUT/chunk_sweep.txt; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/chunk_trace -- python3 tools/prefill_bench.py --kernels auto --rounds 1 --iters 2 --cases $CHUNKS > $OUT/chunk_trace.out 2>&1 || { tail -30 $OUT/chunk_trace.out; exit 1; }
# both prefill kernels forced on every shape of the tool, one process
timeout -k 10 400 python3 tools/prefill_bench.py --kernels 0,1 --rounds 3 --iters 3 --cases cfg2_prefill_7x512,cfg3_chunk_last_8k,cfg3_chunk_mid_4k,cfg3_7seq_chunks,b16_n2048_h32,b4_n8192_h32,b16_n2048_h64 > $OUT/prefill_by_case.txt 2>/dev/null || { tail -5 $OUT/prefill_by_case.txt; exit 1; }
# decode at the one-kv-head shard of config 4: HBM traffic counters (two passes), and the kernel pair of 1 x 32 k
DEC81="--hq 8 --hkv 1 --batch 256 --ctx-lo 2048 --ctx-hi 2048 --layers 8 --rounds 1 --modes auto --chunks 0"
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_h8_1 -- python3 tools/decode_sweep.py $DEC81 > $OUT/pmc_fetch_h8_1.out 2>&1 || { tail -30 $OUT/pmc_fetch_h8_1.out; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_h8_1 -- python3 tools/decode_sweep.py $DEC81 > $OUT/pmc_write_h8_1.out 2>&1 || { tail -30 $OUT/pmc_write_h8_1.out; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dec_1x32k -- python3 tools/decode_sweep.py --hq 32 --hkv 8 --batch 1 --ctx-lo 32768 --ctx-hi 32768 --layers 8 --rounds 3 --modes auto,static --chunks 0 > $OUT/dec_1x32k.out 2>&1 || { tail -30 $OUT/dec_1x32k.out; exit 1; }
find $OUT -name "*kernel_stats.csv" | head -8
echo done
