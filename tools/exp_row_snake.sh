#!/bin/bash
# the headline batch (all its workgroups resident at once): does an order that evens out the keys per compute unit help?
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 9 --modes auto,fused --chunks 0 --layers 8"
for srt in "" desc snake8 snake16 snake32 snake64 snake128; do
  echo -n "B 256 U[513,767] order '${srt:-random}': "; $S ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ chunk= *0 *//' | cut -c1-75 | tr '\n' ';'; echo
done
for srt in "" desc snake32 snake64; do
  echo -n "B 512 U[100,2000] order '${srt:-random}': "; $S --layers 2 --modes auto --batch 512 --ctx-lo 100 --ctx-hi 2000 ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ chunk= *0 *//' | cut -c1-75 | tr '\n' ';'; echo
done
