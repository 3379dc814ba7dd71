#!/bin/bash
# more work items than are resident at once, rows longest first, one long sequence: forced chunks against the plan's own
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0,512,1024 --hint --chunk-by-call --sort desc --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk" | cut -c1-90; }
run --batch 512 --ctx-lo 100 --ctx-hi 2000 --long 1:8000
run --batch 512 --ctx-lo 100 --ctx-hi 2000 --long 1:10000
run --batch 512 --ctx-lo 100 --ctx-hi 2000 --long 1:14000
run --batch 384 --ctx-lo 100 --ctx-hi 2000 --long 1:5000
run --batch 384 --long 1:3000
run --batch 512 --long 1:4000
run --batch 512 --long 1:6000
run --batch 256 --hq 32 --hkv 8 --long 1:3000
run --batch 256 --hq 32 --hkv 8 --ctx-lo 100 --ctx-hi 2000 --long 1:6000
