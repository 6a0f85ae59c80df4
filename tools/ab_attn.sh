#!/bin/bash
# A/B of attention-kernel builds on the GPU box: tools/ab_attn.sh rs0 rs1 ...  (libvcengine_<name>.so; "main" = libvcengine.so)
cd "$(dirname "$0")/.."
for v in "$@"; do
  lib=versecrafter_amd/libvcengine_$v.so
  [ "$v" = main ] && lib=versecrafter_amd/libvcengine.so
  echo "== $v"
  VC_ENGINE_LIB=$PWD/$lib python tools/bench_kernels.py attn attnseg 2>&1 | grep -v "^$"
done
