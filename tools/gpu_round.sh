#!/bin/bash
# One GPU session: parity tests, the bench line, the kernel-trace profile of the same command, PMC passes.
# Every GPU step has its own time limit and the chain stops at the first failure.
set -o pipefail
OUT=gpurun_out/${1:-r02}
BENCH_FLAGS=${2:-}          # e.g. --no-e2e
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -3 $OUT/pytest.log
timeout -k 10 600 python -u bench.py $BENCH_FLAGS > $OUT/bench_line.json 2> $OUT/bench.err || { tail -30 $OUT/bench.err; exit 1; }
cat $OUT/bench_line.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-e2e --no-tp > $OUT/bench_line_under_rocprof.json 2> $OUT/rocprof_stats.err || { tail -30 $OUT/rocprof_stats.err; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-tp --no-prefill --no-configs --attn-only > $OUT/pmc_fetch.out 2>&1 || { tail -30 $OUT/pmc_fetch.out; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-tp --no-prefill --no-configs --attn-only > $OUT/pmc_write.out 2>&1 || { tail -30 $OUT/pmc_write.out; exit 1; }
# the hand-scheduled prefill kernel alone: kernel trace, then MFMA-busy against wave cycles
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prefill_stats -- python3 tools/prefill_bench.py --kernels 1 --cases b4_n8192_h32,b16_n2048_h32 --rounds 2 --iters 3 > $OUT/prefill_stats.out 2>&1 || { tail -30 $OUT/prefill_stats.out; exit 1; }
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_mfma -- python3 tools/prefill_bench.py --kernels 1 --cases b4_n8192_h32 --rounds 1 --iters 2 > $OUT/pmc_mfma.out 2>&1 || { tail -30 $OUT/pmc_mfma.out; exit 1; }
find $OUT -name "*kernel_stats.csv" | head -4
echo done
