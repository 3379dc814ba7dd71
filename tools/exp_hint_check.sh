#!/bin/bash
# the suggestion as built, rows longest first: row "chunk= N" (suggested) against "chunk= 0"; no second row = suggestion 0
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0 --hint --chunk-by-call --sort desc --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk" | cut -c1-90; }
run --batch 64 --long 1:1500
run --batch 64 --long 1:2000
run --batch 64 --ctx-lo 100 --ctx-hi 2000 --long 1:3000
run --batch 256 --long 1:1500
run --batch 256 --ctx-lo 100 --ctx-hi 2000
run --batch 320 --ctx-lo 100 --ctx-hi 2000
run --batch 320 --long 1:3000
run --batch 384 --long 1:3000
run --batch 512 --ctx-lo 100 --ctx-hi 2000 --long 1:6000
run --batch 512 --long 1:8192
run --batch 256 --hq 32 --hkv 8 --ctx-lo 100 --ctx-hi 2000
run --batch 256 --hq 32 --hkv 8 --long 1:3000
run --batch 256 --hq 32 --hkv 8 --long 1:8192
run --batch 128 --hq 32 --hkv 8 --long 1:3000
run --batch 256 --hq 64 --hkv 8 --long 1:8192
