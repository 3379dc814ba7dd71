#!/bin/bash
# Build tuning variants of the library (decode kernel knobs) into nano-vllm_amd/lib/variants/.
# usage: tools/build_variants.sh "name1:-DNV_DEC_WAVES=8" "name2:-DNV_DEC_NT=0" ...
set -e
cd "$(dirname "$0")/../nano-vllm_amd/csrc"
OUT=../lib/variants
mkdir -p $OUT ../../build/variants
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  objs=""
  for f in nv_api store_kvcache norm_rope activation attn_decode attn_prefill attn_prefill_asm; do
    o=../../build/variants/${name}_$f.o
    if [ "$f" = "attn_decode" ] || [ ! -f ../../build/csrc/$f.o ]; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $flags -c $f.hip -o $o &
    else
      cp ../../build/csrc/$f.o $o
    fi
    objs="$objs $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$name.so $objs
  echo "built $OUT/lib_$name.so ($flags)"
done
