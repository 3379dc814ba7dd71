#!/bin/bash
# unsplit launches of many pairs (one wave per pair): the kernel without prefetch (library "default") against one tile of
# prefetch (variant "deep"), by sequence length
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 7 --modes auto --chunks 0 --layers 2"
run() { echo "$*:"; $S "$@" 2>/dev/null | grep "chunk=" | cut -c1-100; }
run --batch 256 --ctx-lo 1024 --ctx-hi 1024
run --batch 256 --ctx-lo 2048 --ctx-hi 2048
run --batch 256 --ctx-lo 768 --ctx-hi 1280
run --batch 256 --ctx-lo 100 --ctx-hi 2000 --sort desc
run --batch 512 --sort desc
run --batch 1024 --sort desc
run --batch 1024 --ctx-lo 100 --ctx-hi 2000 --sort desc
run --batch 512 --ctx-lo 100 --ctx-hi 1000 --sort desc
run --batch 256 --ctx-lo 300 --ctx-hi 400
run --batch 1024 --ctx-lo 100 --ctx-hi 300
