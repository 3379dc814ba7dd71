#!/bin/bash
# Timing-only variants of the hand-scheduled prefill loop (results are wrong by construction): one library
# per ablation under nano-vllm_amd/lib/variants/, then the real loop is regenerated and rebuilt.
set -e
cd "$(dirname "$0")/.."
mkdir -p nano-vllm_amd/lib/variants
for ab in "$@"; do
  PF_ABLATE=$ab python tools/gen_prefill_asm.py > /dev/null
  make -C nano-vllm_amd/csrc -j8 > /dev/null
  cp nano-vllm_amd/lib/libnanovllm_hip.so "nano-vllm_amd/lib/variants/lib_pf_${ab//,/_}.so"
  echo "built $ab"
done
python tools/gen_prefill_asm.py > /dev/null
make -C nano-vllm_amd/csrc -j8 > /dev/null
