#!/bin/bash
# Register budget of the look-ahead: a potrf128 wave must fit on a SIMD (512 VGPRs) beside one wave of the bulk
# trailing update, or the diagonal-block kernel waits for an EMPTY compute unit (see potrf128.h).
# Fails when potrf128_kernel<true> needs more than 264 or a bulk-update kernel more than 248 VGPRs, or when potrf128's LDS
# image and one bulk workgroup's staging buffers together exceed a compute unit's 160 KB.
set -e
cd "$(dirname "$0")/../treegp_amd/csrc"
S=$(mktemp /tmp/chol_XXXX.s)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o $S chol.hip 2>/dev/null
get() { awk -v pat="$1" '$0 ~ "^"pat {f=1} f && /amdhsa_next_free_vgpr/ {print $2; exit}' $S; }
P=$(get "_ZN8potrf_v215potrf128_kernelILb1")
B4=$(get "_ZN12_GLOBAL__N_116syrk_segs_kernelILi4")
B2=$(get "_ZN12_GLOBAL__N_115syrk_dtv_kernelILi2")
D4=$(get "_ZN12_GLOBAL__N_117syrk_distn_kernelILi4")
lds() { awk -v pat="$1" '$0 ~ "^"pat {f=1} f && /amdhsa_group_segment_fixed_size/ {print $2; exit}' $S; }
LP=$(lds "_ZN8potrf_v215potrf128_kernelILb1")
LD=$(lds "_ZN12_GLOBAL__N_117syrk_distn_kernelILi4")
LS=$(lds "_ZN12_GLOBAL__N_116syrk_segs_kernelILi4")
rm -f $S
echo "next_free_vgpr: potrf128_kernel<true> = $P (budget 264); syrk_segs_kernel<4> = $B4, syrk_dtv_kernel<2> = $B2, syrk_distn_kernel<4> = $D4 (budget 248)"
echo "LDS bytes: potrf128 $LP, syrk_distn_kernel<4> $LD, syrk_segs_kernel<4> $LS (potrf128 + one bulk workgroup must fit in 163840)"
[ $((LP + LD)) -le 163840 ] && [ $((LP + LS)) -le 163840 ] && [ "$P" -le 264 ] && [ "$B4" -le 248 ] && [ "$B2" -le 248 ] && [ "$D4" -le 248 ]
