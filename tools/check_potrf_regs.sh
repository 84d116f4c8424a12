#!/bin/bash
# Fails when potrf128_kernel<true> needs more VGPRs than fit beside one trailing-update wave (see potrf128.h).
set -e
cd "$(dirname "$0")/../treegp_amd/csrc"
S=$(mktemp /tmp/chol_XXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o $S chol.hip 2>/dev/null
N=$(awk '/^_ZN8potrf_v215potrf128_kernelILb1/ {f=1} f && /amdhsa_next_free_vgpr/ {print $2; exit}' $S)
rm -f $S
echo "potrf128_kernel<true>: next_free_vgpr = $N (budget 264)"
[ "$N" -le 264 ]
