#!/bin/bash
# spread-out large batches, rows longest first: the suggestion (2048 = unsplit, prefetching kernel) against the plan's own
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 7 --modes auto,fused --chunks 0 --hint --chunk-by-call --sort desc --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk" | cut -c1-100; }
run --batch 384 --ctx-lo 100 --ctx-hi 2000
run --batch 512 --ctx-lo 100 --ctx-hi 2000
run --batch 1024 --ctx-lo 100 --ctx-hi 2000
run --batch 512 --ctx-lo 300 --ctx-hi 1500
