#!/bin/bash
# decode launches with more work items than the chip holds at once: does the order of the sequences matter?
S="timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0 --layers 2"
for B in 512 1024 2048; do for srt in "" desc asc; do
  echo -n "B $B U[100,2000] order '${srt:-random}': "; $S --batch $B --ctx-lo 100 --ctx-hi 2000 ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ *default:auto chunk= *0 *//' | cut -c1-60
done; done
for srt in "" desc; do echo -n "B 1024 U[513,767] order '${srt:-random}': "; $S --batch 1024 --ctx-lo 513 --ctx-hi 767 ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ *default:auto chunk= *0 *//' | cut -c1-60; done
for srt in "" desc asc; do echo -n "B 256 U[513,767] (headline) order '${srt:-random}': "; timeout -k 5 200 python -u tools/decode_sweep.py --rounds 9 --modes auto,fused --chunks 0 --layers 8 ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ chunk= *0 *//' | cut -c1-75 | tr '\n' ';'; echo; done
for srt in "" desc; do echo -n "B 256 U[100,2000] order '${srt:-random}': "; timeout -k 5 200 python -u tools/decode_sweep.py --rounds 5 --modes auto --chunks 0 --layers 4 --ctx-lo 100 --ctx-hi 2000 ${srt:+--sort $srt} 2>/dev/null | grep chunk= | sed 's/ *default:auto chunk= *0 *//' | cut -c1-60; done
