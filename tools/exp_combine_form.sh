#!/bin/bash
# merge launch against in-launch combine (last arriver merges) on the shapes whose launch is short
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 7 --modes auto,single1 --chunks 0 --layers 8"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk=" | cut -c1-100; }
run --hq 8 --hkv 1 --batch 256 --ctx-lo 2048 --ctx-hi 2048
run --hq 16 --hkv 2 --batch 256 --ctx-lo 2048 --ctx-hi 2048
run --hq 32 --hkv 4 --batch 256 --ctx-lo 2048 --ctx-hi 2048
run --hq 32 --hkv 8 --batch 4 --ctx-lo 32768 --ctx-hi 32768
run --hq 32 --hkv 8 --batch 8 --ctx-lo 32768 --ctx-hi 32768
run --hq 16 --hkv 8 --batch 32 --ctx-lo 2048 --ctx-hi 2048
run --hq 16 --hkv 8 --batch 64 --ctx-lo 513 --ctx-hi 767
