#!/bin/sh
# Developer check: prints, per kernel of a .hip file, the loops that LLVM's uniformity analysis sees leaving
# divergently.  In the wave-uniform decode kernels only lane-strided copy loops may show up: anything else means a
# scalar value got merged behind a lane-masked branch and the decoder state moved from SGPRs to VGPRs.
# usage: scripts/check_uniform.sh decompress_amd/csrc/inflate.hip
set -e
f=$1
t=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S -emit-llvm --cuda-device-only -o $t/k.ll "$f" 2>/dev/null
/opt/rocm/lib/llvm/bin/opt -passes='print<uniformity>' -disable-output $t/k.ll 2> $t/u.txt
grep -n "for function\|CYCLES WITH DIVERGENT EXIT\|^  depth=" $t/u.txt | cut -c1-140
echo "(full report: $t/u.txt)"
