#!/bin/bash
# decode launch of the headline shape at the same mean length and growing spread of the lengths
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 7 --modes auto --chunks 0"
for r in "640 640" "632 648" "608 672" "576 704" "544 736" "513 767" "385 895" "257 1023"; do
  set -- $r
  $S --ctx-lo $1 --ctx-hi $2 2>/dev/null | tr '\n' ' '; echo
done
