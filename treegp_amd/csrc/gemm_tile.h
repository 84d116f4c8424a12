// The NT fp64 MFMA tile shared by the Cholesky (chol.hip) and the multi-GPU kernels (dist.hip):
// C (128x128) = or -= A (128 x KDEPTH) * B (128 x KDEPTH)^T, all row-major.
//
// 128x128 per workgroup of 4 waves (2x2, 64x64 per wave = 4x4 v_mfma_f64_16x16x4_f64 tiles, 128
// accumulator VGPRs); operands staged through LDS in 16-deep k-chunks with a row stride of 18
// doubles (conflict-free ds_read_b64 fragment reads: lane (r, kq) -> slot 18 r + kq mod 32 is a
// permutation), double-buffered, the next chunk's global loads in flight during the MFMAs.
// A and C always live in 256-wide panels (ld 256); B is a panel (LDB 256) or a W block (LDB 128).
#pragma once
#include "tgp_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int KB = 16;         // k-chunk depth
constexpr int LS = KB + 2;     // LDS row stride in doubles

// MODE 0: C = A B^T     MODE 1: C -= A B^T      (A: 128 x kdepth, B: 128 x kdepth, row-major)
// A and C always live in 256-wide panels (ld 256); B is a panel (LDB 256) or a W block (LDB 128).
template <int MODE, int LDB, int KDEPTH>
__device__ __forceinline__ void gemm_tile_128(const double *a_ptr, const double *b_ptr, double *c_ptr) {
    constexpr int LDA = TGP_PW, LDC = TGP_PW;
    __shared__ __attribute__((aligned(16))) double lds[2][2][128 * LS];   // [buf][A|B][row*LS + k]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // staging map: piece s of this thread = row (tid>>3) + 32 s, doubles kp..kp+1
    const int srow = tid >> 3;
    const int kp = (tid & 7) * 2;
    const double *ga = a_ptr + srow * LDA + kp;
    const double *gb = b_ptr + srow * LDB + kp;

    double2 ra[4], rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ra[s] = *reinterpret_cast<const double2 *>(ga + s * 32 * LDA);
        rb[s] = *reinterpret_cast<const double2 *>(gb + s * 32 * LDB);
    }

    d4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};

#pragma unroll
    for (int s = 0; s < 4; ++s) {
        *reinterpret_cast<double2 *>(&lds[0][0][(srow + 32 * s) * LS + kp]) = ra[s];
        *reinterpret_cast<double2 *>(&lds[0][1][(srow + 32 * s) * LS + kp]) = rb[s];
    }
    __syncthreads();

    constexpr int nchunk = KDEPTH / KB;
    const int fa = (wr * 64 + l15) * LS + l4;      // fragment read offsets
    const int fb = (wc * 64 + l15) * LS + l4;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        const bool more = (c + 1 < nchunk);
        if (more) {
            const int k0 = (c + 1) * KB;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ra[s] = *reinterpret_cast<const double2 *>(ga + s * 32 * LDA + k0);
                rb[s] = *reinterpret_cast<const double2 *>(gb + s * 32 * LDB + k0);
            }
        }
        const double *As = lds[buf][0];
        const double *Bs = lds[buf][1];
#pragma unroll
        for (int k4 = 0; k4 < KB / 4; ++k4) {
            double af[4], bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = As[fa + m * 16 * LS + k4 * 4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = Bs[fb + n * 16 * LS + k4 * 4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                *reinterpret_cast<double2 *>(&lds[buf ^ 1][0][(srow + 32 * s) * LS + kp]) = ra[s];
                *reinterpret_cast<double2 *>(&lds[buf ^ 1][1][(srow + 32 * s) * LS + kp]) = rb[s];
            }
        }
        __syncthreads();
    }

    // C fragment map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 r
    double *cbase = c_ptr + (wr * 64 + l4) * LDC + wc * 64 + l15;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if constexpr (MODE == 1) {
            double old[4][4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[n][r] = cbase[(m * 16 + 4 * r) * LDC + n * 16];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) cbase[(m * 16 + 4 * r) * LDC + n * 16] = old[n][r] - acc[m][n][r];
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) cbase[(m * 16 + 4 * r) * LDC + n * 16] = acc[m][n][r];
        }
    }
}
