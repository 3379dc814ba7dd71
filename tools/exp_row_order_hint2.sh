#!/bin/bash
# rows longest first: where does the host's length hint (row "chunk= N" against "chunk= 0") still pay?
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0 --hint --chunk-by-call --sort desc --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk" | cut -c1-90; }
run --batch 384 --ctx-lo 100 --ctx-hi 2000
run --batch 512 --ctx-lo 100 --ctx-hi 4000
run --batch 256 --ctx-lo 100 --ctx-hi 4000
run --batch 256 --long 1:8192
run --batch 512 --long 1:8192
run --batch 512 --long 1:16384
run --batch 1024 --long 1:16384
run --batch 1024 --long 1:32768
run --batch 512 --ctx-lo 100 --ctx-hi 1000 --long 8:8192
run --batch 256 --ctx-lo 130 --ctx-hi 383 --long 8:2048
run --batch 256 --hq 32 --hkv 8 --ctx-lo 100 --ctx-hi 2000
run --batch 512 --hq 32 --hkv 8 --ctx-lo 100 --ctx-hi 2000
run --batch 512 --hq 32 --hkv 8 --long 1:16384
