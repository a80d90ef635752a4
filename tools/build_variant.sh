#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]   -> build/exp/lib<name>.so
# Developer A/B builds of the same library (loaded through FLATE_HIP_LIB); never shipped.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
C=moonbit-flate_amd/csrc
mkdir -p build/exp
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Iinclude -I$C "$@" \
  $C/lz77_kernels.hip $C/huff_pack_kernels.hip $C/compact_kernels.hip $C/inflate_kernels.hip \
  $C/splice_kernels.hip $C/flate_api.hip $C/gather.hip $C/checksum.hip $C/synth.cpp -o build/exp/lib$name.so -lpthread -ldl
echo build/exp/lib$name.so
