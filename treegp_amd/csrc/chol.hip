// Blocked right-looking fp64 Cholesky on the packed lower panels (seam S2 of include/tgp.h;
// replaces scipy.linalg.cholesky at treegp/gp_interp.py:181 and log_likelihood.py:30).
//
// Per 256-wide panel k (two 128-column halves):
//   potrf128  diag block 0     one workgroup, in-LDS Gauss-Jordan: L11 in place + W11 = L11^-1
//   gemm<0>   rows below       X = A W11^T            (triangular solve as a GEMM, in place)
//   gemm<1>   column half 1    A[:,128:256] -= X X_d^T (depth 128)
//   potrf128  diag block 1     L22, W22
//   gemm<0>   rows below       X = A W22^T
//   syrk      trailing matrix  C -= P P^T, depth 256, on v_mfma_f64_16x16x4_f64   <- N^3/3 flops
// All GEMMs are the same NT tile: 128x128 per workgroup of 4 waves (2x2, 64x64 per wave =
// 4x4 MFMA tiles, 128 accumulator VGPRs), operands staged through LDS in 16-deep k-chunks
// (row stride 18 doubles -> conflict-free ds_read_b64 fragment reads), double-buffered with
// the next chunk's global loads in flight during the MFMAs; two workgroups per CU so one
// workgroup's C prologue/epilogue hides behind the other's MFMAs.
#include "tgp_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {
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

// a column of 128-row tiles: tile t uses A rows [128 t, +128), the fixed B block, C rows [128 t, +128)
template <int MODE, int LDB>
__global__ __launch_bounds__(256, 2) void gemm_col_kernel(const double *A, const double *B, double *C) {
    const int64_t t = blockIdx.x;
    gemm_tile_128<MODE, LDB, TGP_TB>(A + t * 128 * TGP_PW, B, C + t * 128 * TGP_PW);
}

// trailing update after panel kp: C(ti, tj) -= P_ti P_tj^T over the lower-triangular tile set
__global__ __launch_bounds__(256, 2) void syrk_trailing_kernel(double *Abase, int64_t Np, int kpanel, int T) {
    int ti, tj;
    tilemap(blockIdx.x, T, ti, tj);
    if (ti < 0) return;
    const double *P = Abase + panel_off(kpanel, Np) + (int64_t)TGP_PW * TGP_PW;   // rows below the diag block
    const int64_t pj = kpanel + 1 + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * (kpanel + 1) + (int64_t)TGP_TB * ti;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    gemm_tile_128<1, TGP_PW, TGP_PW>(P + (int64_t)ti * TGP_TB * TGP_PW, P + (int64_t)tj * TGP_TB * TGP_PW, C);
}

// 128x128 diagonal block: Cholesky factor written back in place (lower) and its inverse to W.
// In-LDS Gauss-Jordan: after step j, columns <= j of T hold L^-1 rows, columns > j the Schur
// complement; column j of L goes to global memory as soon as it is final.
__global__ __launch_bounds__(256) void potrf128_kernel(double *A, int lda, double *W, int *info, int base) {
    constexpr int TS = 129;
    __shared__ double T[128 * TS];
    __shared__ double lcol[128], vrow[128];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 128 * 128; idx += 256) {
        const int i = idx >> 7, c = idx & 127;
        T[i * TS + c] = (c <= i) ? A[(int64_t)i * lda + c] : 0.0;
    }
    __syncthreads();
    const int tr = tid >> 4, tc = tid & 15;
    for (int j = 0; j < 128; ++j) {
        const double djj = T[j * TS + j];
        const double d = sqrt(djj);
        const double inv = 1.0 / d;
        if (tid == 0 && !(djj > 0.0)) atomicCAS(info, 0, base + j + 1);
        __syncthreads();                       // everyone has read T[j][j]
        if (tid < 128) {
            const int t = tid;
            if (t > j) {
                const double v = T[t * TS + j] * inv;
                lcol[t] = v;
                vrow[t] = v;
                A[(int64_t)t * lda + j] = v;
            } else if (t == j) {
                vrow[j] = inv;
                T[j * TS + j] = inv;
                A[(int64_t)j * lda + j] = d;
            } else {
                const double v = T[j * TS + t] * inv;
                vrow[t] = v;
                T[j * TS + t] = v;
            }
        }
        __syncthreads();
        for (int i = j + 1 + tr; i < 128; i += 16) {
            const double li = lcol[i];
            for (int c = tc; c <= i; c += 16) {
                const double old = (c == j) ? 0.0 : T[i * TS + c];
                T[i * TS + c] = old - li * vrow[c];
            }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < 128 * 128; idx += 256) {
        const int i = idx >> 7, c = idx & 127;
        W[idx] = (c <= i) ? T[i * TS + c] : 0.0;
    }
}
}  // namespace

int launch_potrf(tgp_ctx *ctx, double *d_A, int64_t Np, double *d_W) {
    TGP_ARG(Np > 0 && Np % TGP_PW == 0);
    hipStream_t st = ctx->stream;
    TGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
    const int nP = (int)(Np / TGP_PW);
    const bool prof = ctx->profiling != 0;
    if (prof) {
        while ((int)ctx->prof_events.size() < 2 * nP) {
            hipEvent_t e;
            TGP_HIP(hipEventCreate(&e));
            ctx->prof_events.push_back(e);
        }
    }
    double flops = 0.0;
    int nlaunch = 0;
    for (int k = 0; k < nP; ++k) {
        double *Pk = d_A + panel_off(k, Np);
        const int64_t mk = Np - (int64_t)TGP_PW * k;
        double *W0 = d_W + (int64_t)(2 * k) * TGP_TB * TGP_TB;
        double *W1 = W0 + TGP_TB * TGP_TB;
        double *R1 = Pk + (int64_t)TGP_TB * TGP_PW;                 // row 128 of the panel
        const int r1 = (int)((mk - TGP_TB) / TGP_TB);
        potrf128_kernel<<<1, 256, 0, st>>>(Pk, TGP_PW, W0, ctx->d_info, (int)(k * TGP_PW));
        gemm_col_kernel<0, TGP_TB><<<r1, 256, 0, st>>>(R1, W0, R1);
        gemm_col_kernel<1, TGP_PW><<<r1, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
        potrf128_kernel<<<1, 256, 0, st>>>(R1 + TGP_TB, TGP_PW, W1, ctx->d_info, (int)(k * TGP_PW + TGP_TB));
        const int r2 = (int)((mk - TGP_PW) / TGP_TB);
        if (r2 > 0) {
            double *R2 = Pk + (int64_t)TGP_PW * TGP_PW + TGP_TB;    // row 256, column 128
            gemm_col_kernel<0, TGP_TB><<<r2, 256, 0, st>>>(R2, W1, R2);
            if (prof) TGP_HIP(hipEventRecord(ctx->prof_events[2 * nlaunch], st));
            syrk_trailing_kernel<<<(unsigned)tilemap_grid(r2), 256, 0, st>>>(d_A, Np, k, r2);
            if (prof) TGP_HIP(hipEventRecord(ctx->prof_events[2 * nlaunch + 1], st));
            { const double m = (double)r2 * TGP_TB; flops += (double)TGP_PW * m * (m + 1.0); }   // algorithmic: lower triangle only
            ++nlaunch;
        }
    }
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    ctx->timings[6] = nlaunch;
    ctx->timings[7] = flops;
    ctx->timings[5] = 0.0;
    if (prof) {
        double tot = 0.0;
        for (int i = 0; i < nlaunch; ++i) {
            float ms = 0.f;
            TGP_HIP(hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
            tot += ms;
        }
        ctx->timings[5] = tot;
    }
    return *ctx->h_info;
}

int tgp_debug_tilemap(int64_t T, int32_t *ti, int32_t *tj, int64_t cap) {
    const int64_t g = tilemap_grid(T);
    if (cap < g) return -1;
    for (int64_t b = 0; b < g; ++b) {
        int i, j;
        tilemap(b, T, i, j);
        ti[b] = i;
        tj[b] = j;
    }
    return (int)g;
}
