for heads in "8 1" "16 8" "32 8"; do
  set -- $heads
  for ctx in 1024 4096 16384; do
    echo "== H $1/$2 B 1 ctx $ctx"
    timeout -k 5 120 python -u tools/decode_sweep.py --rounds 5 --modes auto,static,single1 --chunks 0,64,128 --layers 4 --hq $1 --hkv $2 --batch 1 --ctx-lo $ctx --ctx-hi $ctx 2>/dev/null | grep "chunk=" | cut -c1-75
  done
done
