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

#include "gemm_tile.h"
#include "potrf128.h"

namespace {
// a column of 128-row tiles: tile t uses A rows [128 t, +128), the fixed B block, C rows [128 t, +128)
template <int MODE, int LDB>
__global__ __launch_bounds__(256, 2) void gemm_col_kernel(const double *A, const double *B, double *C) {
    const int64_t t = blockIdx.x;
    gemm_tile_128<MODE, LDB, TGP_TB>(A + t * 128 * TGP_PW, B, C + t * 128 * TGP_PW);
}

// trailing update after panel kp: C(ti, tj) -= P_ti P_tj^T over the lower-triangular tile set
template <typename CFG, int STAGGER>
__global__ __launch_bounds__(256, 2) void syrk_trailing_kernel(double *Abase, int64_t Np, int kpanel, int T,
                                                               int64_t nvirt, unsigned long long *stamps = nullptr) {
    // persistent when gridDim.x < nvirt: each workgroup walks the virtual block ids b, b + gridDim.x, ...
    // (gridDim.x is a multiple of 8, so a workgroup keeps its XCD's share of the super-tiles)
    const double *P = Abase + panel_off(kpanel, Np) + (int64_t)TGP_PW * TGP_PW;   // rows below the diag block
    if constexpr (STAGGER > 0) {
        if (blockIdx.x >= 256 && blockIdx.x < 512) {
#pragma unroll 1
            for (int i = 0; i < STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
        }
    }
#pragma unroll 1
    for (int64_t vb = blockIdx.x; vb < nvirt; vb += gridDim.x) {
        int ti, tj;
        tilemap(vb, T, ti, tj);
        if (ti < 0) continue;
        const int64_t pj = kpanel + 1 + (tj >> 1);
        const int64_t I = (int64_t)TGP_PW * (kpanel + 1) + (int64_t)TGP_TB * ti;
        double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
        gemm_tile_128<1, TGP_PW, TGP_PW, CFG>(P + (int64_t)ti * TGP_TB * TGP_PW, P + (int64_t)tj * TGP_TB * TGP_PW, C,
                                              stamps ? stamps + 4 * vb : nullptr);
    }
}

// First-generation 128x128 diagonal-block kernel (kept for A/B runs, TGP_POTRF_VARIANT=0; the
// default is potrf_v2::potrf128_kernel in potrf128.h).  In-LDS Gauss-Jordan: after step j, columns <= j of T hold L^-1 rows, columns > j the Schur
// complement; column j of L goes to global memory as soon as it is final.
__global__ __launch_bounds__(256) void potrf128_lds_kernel(double *A, int lda, double *W, int *info, int base) {
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

// multi-GPU trailing update: rank g updates its own block rows.  P is the all-gathered panel,
// laid out [rank][cmax blocks][256][256]; blockIdx.y = local tile row among blocks > k,
// blockIdx.x = trailing tile column (tiles right of the diagonal exit).
__global__ __launch_bounds__(256, 2) void syrk_dist_kernel(double *Aloc, const int64_t *__restrict__ loff, int kpanel,
                                                           int G, int g, int cmax, const double *P) {
    const int lt = blockIdx.y;
    const int64_t gtj = blockIdx.x;
    const int64_t s0 = kpanel + 1;
    const int64_t bi = dist_first_ge(s0, g, G) + (int64_t)(lt >> 1) * G;
    const int64_t gti = 2 * (bi - s0) + (lt & 1);
    if (gtj > gti) return;
    const int64_t bj = s0 + (gtj >> 1);
    const int rj = (int)(bj % G);
    const int64_t idxj = (bj - dist_first_ge(s0, rj, G)) / G;
    const double *a = P + (((int64_t)g * cmax + (lt >> 1)) * TGP_PW + (lt & 1) * TGP_TB) * TGP_PW;
    const double *b = P + (((int64_t)rj * cmax + idxj) * TGP_PW + (gtj & 1) * TGP_TB) * TGP_PW;
    double *c = Aloc + loff[bj] + (((bi - dist_first_ge(bj, g, G)) / G) * TGP_PW + (lt & 1) * TGP_TB) * TGP_PW +
                (gtj & 1) * TGP_TB;
    gemm_tile_128<1, TGP_PW, TGP_PW>(a, b, c);
}

inline void run_potrf128(hipStream_t st, double *A, int lda, double *W, int *info, int base) {
    static const int variant = [] { const char *e = getenv("TGP_POTRF_VARIANT"); return e ? atoi(e) : 1; }();
    if (variant == 0) potrf128_lds_kernel<<<1, 256, 0, st>>>(A, lda, W, info, base);
    else potrf_v2::potrf128_kernel<<<1, 256, 0, st>>>(A, lda, W, info, base);
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
    const char *sv = getenv("TGP_SYRK_VARIANT");
    const int syrk_variant = sv ? atoi(sv) : 2;
    const char *pv = getenv("TGP_SYRK_PERSIST");
    const int persist = pv ? atoi(pv) : 0;          // workgroups of the persistent grid (0 = one per tile)
    for (int k = 0; k < nP; ++k) {
        double *Pk = d_A + panel_off(k, Np);
        const int64_t mk = Np - (int64_t)TGP_PW * k;
        double *W0 = d_W + (int64_t)(2 * k) * TGP_TB * TGP_TB;
        double *W1 = W0 + TGP_TB * TGP_TB;
        double *R1 = Pk + (int64_t)TGP_TB * TGP_PW;                 // row 128 of the panel
        const int r1 = (int)((mk - TGP_TB) / TGP_TB);
        run_potrf128(st, Pk, TGP_PW, W0, ctx->d_info, (int)(k * TGP_PW));
        gemm_col_kernel<0, TGP_TB><<<r1, 256, 0, st>>>(R1, W0, R1);
        gemm_col_kernel<1, TGP_PW><<<r1, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
        run_potrf128(st, R1 + TGP_TB, TGP_PW, W1, ctx->d_info, (int)(k * TGP_PW + TGP_TB));
        const int r2 = (int)((mk - TGP_PW) / TGP_TB);
        if (r2 > 0) {
            double *R2 = Pk + (int64_t)TGP_PW * TGP_PW + TGP_TB;    // row 256, column 128
            gemm_col_kernel<0, TGP_TB><<<r2, 256, 0, st>>>(R2, W1, R2);
            if (prof) TGP_HIP(hipEventRecord(ctx->prof_events[2 * nlaunch], st));
            const int64_t nvirt = tilemap_grid(r2);
            unsigned gs = (unsigned)nvirt;
            if (persist > 0 && nvirt > persist) gs = (unsigned)persist;
            switch (syrk_variant) {
                case 0: syrk_trailing_kernel<TileCfg<18, false>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 1: syrk_trailing_kernel<TileCfg<17, false>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 2: syrk_trailing_kernel<TileCfg<17, true>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 3: syrk_trailing_kernel<TileCfg<17, true>, 8><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 4: syrk_trailing_kernel<TileCfg<17, true>, 4><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 5: syrk_trailing_kernel<TileCfg<18, true>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 18: syrk_trailing_kernel<TileCfg<17, true, 8>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 19: syrk_trailing_kernel<TileCfg<17, true, 16>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 20: syrk_trailing_kernel<TileCfg<17, true, 24>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 11: syrk_trailing_kernel<TileCfg<17, true, 1>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 12: syrk_trailing_kernel<TileCfg<17, true, 2>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 16: syrk_trailing_kernel<TileCfg<17, true, 6>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 17: syrk_trailing_kernel<TileCfg<17, true, 7>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
                case 30: {       // development: in-kernel stamps of the first (largest) launch
                    static unsigned long long *d_st = nullptr;
                    if (k == 0) {
                        if (!d_st) (void)hipMalloc((void **)&d_st, (size_t)nvirt * 32);
                        (void)hipMemsetAsync(d_st, 0, (size_t)nvirt * 32, st);
                        syrk_trailing_kernel<TileCfg<17, true>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt, d_st);
                        std::vector<unsigned long long> h((size_t)nvirt * 4);
                        (void)hipMemcpyAsync(h.data(), d_st, (size_t)nvirt * 32, hipMemcpyDeviceToHost, st);
                        (void)hipStreamSynchronize(st);
                        double a = 0, b = 0, e = 0, rt = 0; long cnt = 0;
                        for (int64_t i = 0; i < nvirt; ++i) if (h[4 * i + 1]) { a += h[4 * i]; b += h[4 * i + 1]; e += h[4 * i + 2]; rt += h[4 * i + 3]; ++cnt; }
                        fprintf(stderr, "[stamps] tiles %ld  prologue %.0f  loop %.0f  epilogue %.0f cycles (mean); loop clock %.3f GHz\n",
                                cnt, a / cnt, b / cnt, e / cnt, b / rt * 0.1);
                    } else {
                        syrk_trailing_kernel<TileCfg<17, true>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt);
                    }
                } break;
                default: syrk_trailing_kernel<TileCfg<17, true>, 0><<<gs, 256, 0, st>>>(d_A, Np, k, r2, nvirt); break;
            }
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


// 256x256 diagonal block (ld 256): L in place, inverses of its two 128-blocks to W0 / W1
int launch_factor_diag256(tgp_ctx *ctx, double *blk, double *W0, double *W1, int base) {
    hipStream_t st = ctx->stream;
    double *R1 = blk + (int64_t)TGP_TB * TGP_PW;
    run_potrf128(st, blk, TGP_PW, W0, ctx->d_info, base);
    gemm_col_kernel<0, TGP_TB><<<1, 256, 0, st>>>(R1, W0, R1);
    gemm_col_kernel<1, TGP_PW><<<1, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
    run_potrf128(st, R1 + TGP_TB, TGP_PW, W1, ctx->d_info, base + TGP_TB);
    TGP_HIP(hipGetLastError());
    return 0;
}

// rows (ntiles x 128, ld 256) <- rows L_kk^-T with L_kk given by its 256x256 block and W0, W1
int launch_trsm_rows(tgp_ctx *ctx, double *rows, int ntiles, const double *Lkk, const double *W0, const double *W1) {
    if (ntiles <= 0) return 0;
    hipStream_t st = ctx->stream;
    gemm_col_kernel<0, TGP_TB><<<ntiles, 256, 0, st>>>(rows, W0, rows);
    gemm_col_kernel<1, TGP_PW><<<ntiles, 256, 0, st>>>(rows, Lkk + (int64_t)TGP_TB * TGP_PW, rows + TGP_TB);
    gemm_col_kernel<0, TGP_TB><<<ntiles, 256, 0, st>>>(rows + TGP_TB, W1, rows + TGP_TB);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_syrk_dist(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                     const double *d_P, int cmax) {
    const int64_t nB = Np / TGP_PW;
    const int64_t nloc = dist_panel_blocks(kpanel + 1, nB, g, G);     // local blocks > k
    const int64_t ncol = 2 * (nB - kpanel - 1);
    if (nloc <= 0 || ncol <= 0) return 0;
    dim3 grid((unsigned)ncol, (unsigned)(2 * nloc));
    syrk_dist_kernel<<<grid, 256, 0, ctx->stream>>>(d_Aloc, d_loff, kpanel, G, g, cmax, d_P);
    TGP_HIP(hipGetLastError());
    return 0;
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
