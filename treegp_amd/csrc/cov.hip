// Posterior covariance (seam S3b): cov = k(Xs, Xs) - HT (K + D)^-1 HT^T  (treegp/gp_interp.py:184-192)
// from the factor kept by tgp_gp_solve.  With Bt = HT L^-T (M x N) the result is
// Kss - Bt Bt^T, which reuses the factor instead of factorising a second time (the reference's
// own comment at gp_interp.py:189) and keeps the subtraction symmetric.
//
// Everything is the tuned NT MFMA tile of gemm_tile.h, so Bt and the covariance live in the same
// 256-wide panel layout as the factor (rows = query points, Mp x 256 per panel, ld 256):
//   right-looking block substitution over the 128-column blocks kb of L
//     Bt[:, kb]  = Bt[:, kb] W_kb^T                      (W_kb = inverse of the diagonal block)
//     Bt[:, c]  -= Bt[:, kb] L[c, kb]^T    for c > kb
//   cov = Kss - Bt Bt^T, one pass over all panels of Bt (run-time segment loop of the tile).
#include "tgp_internal.h"
#include "kernel_eval.h"
#include "gemm_tile.h"

namespace {
// out: panels of 256 columns, panel p at out + p * rows * 256, element (i, j) -> [i][j & 255]
template <int KE>
__global__ __launch_bounds__(256) void cross_panels_kernel(KParams p, const double *__restrict__ Xs, int64_t m,
                                                           const double *__restrict__ X, int64_t n, int self,
                                                           double *__restrict__ out, int64_t rows, int64_t ncols) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= ncols) return;
    double v = 0.0;
    if (i < m && j < n) {
        v = kernel_value<KE>(p, Xs[2 * i] - X[2 * j], Xs[2 * i + 1] - X[2 * j + 1]);
        if (self && i == j) v = p.amp;
    }
    out[(j >> 8) * rows * TGP_PW + i * TGP_PW + (j & 255)] = v;
}

int launch_cross_panels(tgp_ctx *ctx, const tgp_kernel *k, const double *d_Xs, int64_t m, const double *d_X, int64_t n,
                        int self, double *d_out, int64_t rows, int64_t ncols) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0 && rows <= 65535);
    const KParams p = make_kparams(k);
    dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)rows), block(256);
    switch (ke) {
        case KE_GAUSS: cross_panels_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, rows, ncols); break;
        case KE_VK: cross_panels_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, rows, ncols); break;
        default: cross_panels_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, rows, ncols); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

// Bt[:, kb] <- Bt[:, kb] W_kb^T      (one 128-row tile per workgroup)
__global__ __launch_bounds__(256, 2) void cov_trsm_kernel(double *Bk, const double *W) {
    const int64_t t = blockIdx.x;
    gemm_tile_128<0, TGP_TB, TGP_TB>(Bk + t * TGP_TB * TGP_PW, W, Bk + t * TGP_TB * TGP_PW);
}

// Bt[:, c] -= Bt[:, kb] L[c, kb]^T for c = kb + 1 + blockIdx.y
__global__ __launch_bounds__(256, 2) void cov_update_kernel(double *Bt, int64_t Mp, const double *A, int64_t Np, int kb) {
    const int64_t ti = blockIdx.x;
    const int64_t c = kb + 1 + blockIdx.y;
    const int64_t p = kb >> 1;
    const double *a = Bt + p * Mp * TGP_PW + ti * TGP_TB * TGP_PW + (kb & 1) * TGP_TB;
    const double *b = A + panel_off(p, Np) + (c * TGP_TB - p * TGP_PW) * TGP_PW + (kb & 1) * TGP_TB;
    double *cc = Bt + (c >> 1) * Mp * TGP_PW + ti * TGP_TB * TGP_PW + (c & 1) * TGP_TB;
    gemm_tile_dtv<4, TGP_TB, 1>(a, b, cc, nullptr, nullptr);
}

// ---- the substitution in steps of S = 1024 (512) columns, with the factor's inverse slabs (trsv_big.hip) -------------------------
//   T  = Bt[:, K] V_K^T            (V_K = inverse of the S x S diagonal block: k <= j, 256-deep segments; out of place)
//   Bt[:, c] -= Bt[:, K] L[c, K]^T  for the tile columns c right of super-block K, ONE pass of depth 1024 on the DTV tile
// instead of eight dependent pairs of depth-128 launches per super-block.
__global__ __launch_bounds__(256) void cov_diag_big_kernel(const double *__restrict__ Bt, int64_t Mp, const double *__restrict__ V,
                                                           int64_t Np, int64_t r0, int ncolt, double *__restrict__ T) {
    const int ti = blockIdx.x / ncolt, tj = blockIdx.x % ncolt;          // 128-row tile of the queries, 128-column tile of the block
    const int64_t p0 = r0 >> 8;
    const double *a = Bt + p0 * Mp * TGP_PW + (int64_t)ti * TGP_TB * TGP_PW;
    const double *b = V + (r0 + (int64_t)tj * TGP_TB) * TGP_PW;          // slab panel 0, rows r0 + 128 tj ..
    double *c = T + (int64_t)(tj >> 1) * Mp * TGP_PW + (int64_t)ti * TGP_TB * TGP_PW + (tj & 1) * TGP_TB;
    const int nseg = (tj >> 1) + 1;                                      // V_K is lower triangular: columns k <= j
    gemm_tile_128<0, TGP_PW, TGP_PW, TileDefault, 0>(a, b, c, nullptr, nullptr, nseg, Mp * TGP_PW, Np * TGP_PW);
}

template <int NS>                                                          // NS = S / 256 panels per super-block
__global__ __launch_bounds__(256, 2) void cov_update_big_kernel(double *Bt, int64_t Mp, const double *A, int64_t Np, int64_t r0) {
    const int64_t ti = blockIdx.x;
    const int64_t p0 = r0 >> 8;
    const int64_t c = (r0 >> 7) + 2 * NS + blockIdx.y;                   // global 128-tile column right of the super-block
    SegPtrs<NS> sp;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        sp.a[s] = Bt + (p0 + s) * Mp * TGP_PW + ti * TGP_TB * TGP_PW;
        sp.b[s] = A + panel_off(p0 + s, Np) + (c * TGP_TB - (p0 + s) * TGP_PW) * TGP_PW;
    }
    double *cc = Bt + (c >> 1) * Mp * TGP_PW + ti * TGP_TB * TGP_PW + (c & 1) * TGP_TB;
    gemm_tile_dtv_segs<4, TGP_PW, NS>(sp, cc);
}

// C(ti, tj) -= sum over all panels of Bt[ti] Bt[tj]^T        (C in the same panel layout, Mp rows)
__global__ __launch_bounds__(256, 2) void cov_syrk_kernel(double *Cpm, const double *Bt, int64_t Mp, int nP) {
    const int64_t ti = blockIdx.x, tj = blockIdx.y;
    double *c = Cpm + (tj >> 1) * Mp * TGP_PW + ti * TGP_TB * TGP_PW + (tj & 1) * TGP_TB;
    gemm_tile_dtv<4, TGP_PW, 0>(Bt + ti * TGP_TB * TGP_PW, Bt + tj * TGP_TB * TGP_PW, c, nullptr, nullptr, nP, Mp * TGP_PW,
                                Mp * TGP_PW);
}
}  // namespace

namespace {
// geometry + scratch shared by the two entry points
struct CovPlan {
    int64_t n, m, Np, Mp;
    int nP, nPm;
    double *d_X, *d_Xs, *d_Bt, *d_C;
};
int cov_plan(tgp_ctx *ctx, const tgp_factor *f, int64_t m, bool coords, CovPlan *pl) {
    pl->n = f->n; pl->m = m; pl->Np = f->Np;
    pl->nP = (int)(pl->Np / TGP_PW);
    pl->Mp = (m + TGP_PW - 1) / TGP_PW * TGP_PW;                  // 256: the covariance uses the panel layout too
    pl->nPm = (int)(pl->Mp / TGP_PW);
    TGP_ARG(pl->Mp <= 65535);
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = (coords ? rup(2 * pl->n * 8) + rup(2 * m * 8) : 0) + rup((size_t)pl->Mp * pl->Np * 8) + rup((size_t)pl->Mp * pl->Mp * 8);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return (double *)p; };
    pl->d_X = coords ? take(2 * pl->n * 8) : nullptr;
    pl->d_Xs = coords ? take(2 * m * 8) : nullptr;
    pl->d_Bt = take((size_t)pl->Mp * pl->Np * 8);
    pl->d_C = take((size_t)pl->Mp * pl->Mp * 8);
    return 0;
}
// Bt <- Bt L^-T by block substitution.  `tri`: Bt starts as the identity (m == n), so at the step that eliminates columns
// [r0, r0 + rows) only the row tiles above r0 + rows hold anything: the launches cover those tiles only (a third of the work).
int cov_substitute(tgp_ctx *ctx, tgp_factor *f, const CovPlan &pl, bool tri) {
    hipStream_t st = ctx->stream;
    const int nb = 2 * pl.nP;
    const unsigned mt = (unsigned)(pl.Mp / TGP_TB);
    const bool no_big = getenv("TGP_COV_BIG") && atoi(getenv("TGP_COV_BIG")) == 0;     // A/B: the 128-block substitution
    int S = 0;
    const double *slabs = nullptr;
    if (!no_big) {
        int rc = factor_slabs(ctx, f, 1024, &S, &slabs);
        if (rc) return rc;
    }
    if ((S == 1024 || S == 512) && slabs) {
        int rc = tgp_ensure_scratch2(ctx, (size_t)pl.Mp * S * sizeof(double));
        if (rc) return rc;
        double *T = (double *)ctx->scratch2;
        for (int64_t r0 = 0; r0 < pl.Np; r0 += S) {
            const int64_t rows = (pl.Np - r0) < S ? (pl.Np - r0) : S;
            const int ncolt = (int)(rows / TGP_TB);
            const unsigned mte = tri ? (unsigned)((r0 + rows) / TGP_TB) : mt;          // row tiles that are not all zero yet
            cov_diag_big_kernel<<<mte * (unsigned)ncolt, 256, 0, st>>>(pl.d_Bt, pl.Mp, slabs, pl.Np, r0, ncolt, T);
            for (int64_t q = 0; q < rows / TGP_PW; ++q)
                TGP_HIP(hipMemcpyAsync(pl.d_Bt + ((r0 >> 8) + q) * pl.Mp * TGP_PW, T + q * pl.Mp * TGP_PW,
                                       (size_t)mte * TGP_TB * TGP_PW * sizeof(double), hipMemcpyDeviceToDevice, st));
            const int64_t right = (pl.Np - (r0 + rows)) / TGP_TB;
            if (right > 0 && S == 1024) cov_update_big_kernel<4><<<dim3(mte, (unsigned)right), 256, 0, st>>>(pl.d_Bt, pl.Mp, f->d_A, pl.Np, r0);
            if (right > 0 && S == 512) cov_update_big_kernel<2><<<dim3(mte, (unsigned)right), 256, 0, st>>>(pl.d_Bt, pl.Mp, f->d_A, pl.Np, r0);
        }
    } else {
        for (int kb = 0; kb < nb; ++kb) {
            double *Bk = pl.d_Bt + (int64_t)(kb >> 1) * pl.Mp * TGP_PW + (kb & 1) * TGP_TB;
            const unsigned mte = tri ? (unsigned)(kb + 1) : mt;
            cov_trsm_kernel<<<mte, 256, 0, st>>>(Bk, f->d_W + (int64_t)kb * TGP_TB * TGP_TB);
            const int nc = nb - kb - 1;
            if (nc > 0) cov_update_kernel<<<dim3(mte, (unsigned)nc), 256, 0, st>>>(pl.d_Bt, pl.Mp, f->d_A, pl.Np, kb);
        }
    }
    TGP_HIP(hipGetLastError());
    return 0;
}
// d_Bt holds HT, d_C holds k(X2, X2), both in zero-padded panels: substitution, Kss - Bt Bt^T, result to the host
int cov_finish(tgp_ctx *ctx, tgp_factor *f, const CovPlan &pl, double *cov) {
    hipStream_t st = ctx->stream;
    const unsigned mt = (unsigned)(pl.Mp / TGP_TB);
    int rc = cov_substitute(ctx, f, pl, false);
    if (rc) return rc;
    cov_syrk_kernel<<<dim3(mt, mt), 256, 0, st>>>(pl.d_C, pl.d_Bt, pl.Mp, pl.nP);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    for (int p = 0; p < pl.nPm; ++p) {
        const int64_t w = (pl.m - (int64_t)p * TGP_PW < TGP_PW) ? pl.m - (int64_t)p * TGP_PW : TGP_PW;
        if (w <= 0) break;
        TGP_HIP(hipMemcpy2DAsync(cov + (int64_t)p * TGP_PW, (size_t)pl.m * 8, pl.d_C + (int64_t)p * pl.Mp * TGP_PW, (size_t)TGP_PW * 8,
                                 (size_t)w * 8, (size_t)pl.m, hipMemcpyDeviceToHost, st));
    }
    TGP_HIP(hipEventRecord(ctx->ev[2], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[3] = ms;                       // device compute
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    ctx->timings[9] = ms;                       // (m, m) result to the caller's buffer
    return 0;
}
}  // namespace

extern "C" int tgp_gp_predict_cov(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *X, int64_t n,
                                  const double *Xs, int64_t m, double *cov) {
    TGP_ARG(f && k && X && Xs && cov && m > 0 && n == f->n);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    CovPlan pl;
    int rc = cov_plan(ctx, f, m, true, &pl);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(pl.d_X, X, 2 * n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(pl.d_Xs, Xs, 2 * m * 8, hipMemcpyHostToDevice, st));
    // HT (gp_interp.py:177) and k(X2) (gp_interp.py:191), zero padded, in panels
    rc = launch_cross_panels(ctx, k, pl.d_Xs, m, pl.d_X, n, 0, pl.d_Bt, pl.Mp, pl.Np);
    if (rc) return rc;
    rc = launch_cross_panels(ctx, k, pl.d_Xs, m, pl.d_Xs, m, 1, pl.d_C, pl.Mp, pl.Mp);
    if (rc) return rc;
    return cov_finish(ctx, f, pl, cov);
}

// The same with HT = kernel(X2, Y=X1) (m, n) and Kss = kernel(X2) (m, m) evaluated by the caller (row-major host arrays):
// any scikit-learn kernel tree (gp_interp.py:184-192 with the factor kept by tgp_gp_solve_dense).
extern "C" int tgp_gp_predict_cov_dense(tgp_ctx *ctx, tgp_factor *f, const double *HT, const double *Kss, int64_t m, double *cov) {
    TGP_ARG(f && HT && Kss && cov && m > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    CovPlan pl;
    int rc = cov_plan(ctx, f, m, false, &pl);
    if (rc) return rc;
    const int64_t n = f->n;
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemsetAsync(pl.d_Bt, 0, (size_t)pl.Mp * pl.Np * 8, st));
    TGP_HIP(hipMemsetAsync(pl.d_C, 0, (size_t)pl.Mp * pl.Mp * 8, st));
    for (int p = 0; p < pl.nP; ++p) {
        const int64_t w = (n - (int64_t)p * TGP_PW < TGP_PW) ? n - (int64_t)p * TGP_PW : TGP_PW;
        if (w <= 0) break;
        TGP_HIP(hipMemcpy2DAsync(pl.d_Bt + (int64_t)p * pl.Mp * TGP_PW, (size_t)TGP_PW * 8, HT + (int64_t)p * TGP_PW, (size_t)n * 8,
                                 (size_t)w * 8, (size_t)m, hipMemcpyHostToDevice, st));
    }
    for (int p = 0; p < pl.nPm; ++p) {
        const int64_t w = (m - (int64_t)p * TGP_PW < TGP_PW) ? m - (int64_t)p * TGP_PW : TGP_PW;
        if (w <= 0) break;
        TGP_HIP(hipMemcpy2DAsync(pl.d_C + (int64_t)p * pl.Mp * TGP_PW, (size_t)TGP_PW * 8, Kss + (int64_t)p * TGP_PW, (size_t)m * 8,
                                 (size_t)w * 8, (size_t)m, hipMemcpyHostToDevice, st));
    }
    return cov_finish(ctx, f, pl, cov);
}

// ---- gradient of the log marginal likelihood (kernel derivative convention of treegp/kernels.py:128-150) ------------
//   dlogL/dp = 1/2 sum_ij (alpha_i alpha_j - [K^-1]_ij) dK_ij/dp
// for the four numbers a Gaussian kernel is made of on the device: p = log amp, a, b, c (invLam 00, 01 = 10, 11); the chain rule
// from there to theta stays on the host (kernels.spec_jacobian: the dInvLam/dtheta matrices of kernels.py:138-145).
// K^-1 = L^-T L^-1 comes from the substitution above started at the identity: Bt = L^-T is upper triangular, so both the
// substitution and the product Bt Bt^T skip what is known to be zero (2/3 N^3 flops together, twice a factorisation), and the
// sum over (i, j) is one pass over the lower triangle with dK/dp evaluated from the coordinates.
namespace {
__global__ __launch_bounds__(256) void ident_panels_kernel(double *Bt, int64_t Mp, int64_t Np) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Np) Bt[(i >> 8) * Mp * TGP_PW + i * TGP_PW + (i & 255)] = 1.0;
}

// lower tile pairs (ti >= tj), linear in blockIdx.x:  C(ti, tj) = -sum_{p >= ti / 2} Bt[ti][p] Bt[tj][p]^T   (C zero before)
__global__ __launch_bounds__(256, 2) void kinv_syrk_kernel(double *Cpm, const double *Bt, int64_t Mp, int nP) {
    const int64_t t = blockIdx.x;
    int64_t ti = (int64_t)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > t) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    const int64_t tj = t - ti * (ti + 1) / 2;
    const int64_t p0 = ti >> 1;                                     // row tile ti of L^-T is zero left of its own panel
    const double *a = Bt + p0 * Mp * TGP_PW + ti * TGP_TB * TGP_PW;
    const double *b = Bt + p0 * Mp * TGP_PW + tj * TGP_TB * TGP_PW;
    double *c = Cpm + (tj >> 1) * Mp * TGP_PW + ti * TGP_TB * TGP_PW + (tj & 1) * TGP_TB;
    gemm_tile_dtv<4, TGP_PW, 0>(a, b, c, nullptr, nullptr, nP - (int)p0, Mp * TGP_PW, Mp * TGP_PW);
}

// 64 rows x one 256-column panel per workgroup over the lower triangle; four partial sums per workgroup
__global__ __launch_bounds__(256) void loglik_grad_kernel(KParams p, const double *__restrict__ X, const double *__restrict__ alpha,
                                                          const double *__restrict__ Cpm, int64_t Mp, int64_t n,
                                                          double *__restrict__ partial) {
    __shared__ double red[4][4];
    const int tid = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * TGP_PW + tid;
    const int64_t i0 = (int64_t)blockIdx.y * 64;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    if ((int64_t)blockIdx.x * TGP_PW <= i0 + 63 && j < n) {
        const double xj = X[2 * j], yj = X[2 * j + 1], aj = alpha[j];
        const double *col = Cpm + (int64_t)blockIdx.x * Mp * TGP_PW + tid;
        for (int r = 0; r < 64; ++r) {
            const int64_t i = i0 + r;
            if (i >= n) break;
            if (j > i) continue;
            const double m = alpha[i] * aj + col[i * TGP_PW];       // alpha_i alpha_j - [K^-1]_ij   (C holds -K^-1)
            if (i == j) {
                acc[0] += 0.5 * m * p.amp;                          // the pair (i, i) counts once, d K_ii / d log amp = amp
            } else {
                const double dx = X[2 * i] - xj, dy = X[2 * i + 1] - yj;
                const double e = p.amp * exp(-0.5 * quad_form(p, dx, dy)) * m;
                acc[0] += e;
                acc[1] -= 0.5 * e * dx * dx;
                acc[2] -= e * dx * dy;
                acc[3] -= 0.5 * e * dy * dy;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if ((tid & 63) == 0) red[tid >> 6][q] = v;
    }
    __syncthreads();
    if (tid < 4) partial[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// fixed-order sum of the workgroups' partial sums (one workgroup: the result does not depend on the schedule)
__global__ __launch_bounds__(256) void loglik_grad_reduce_kernel(const double *__restrict__ partial, int64_t count, double *__restrict__ out) {
    __shared__ double red[256][4];
    const int tid = threadIdx.x;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t q = tid; q < count; q += 256)
        for (int s = 0; s < 4; ++s) acc[s] += partial[q * 4 + s];
    for (int s = 0; s < 4; ++s) red[tid][s] = acc[s];
    __syncthreads();
    for (int step = 128; step > 0; step >>= 1) {
        if (tid < step)
            for (int s = 0; s < 4; ++s) red[tid][s] += red[tid + step][s];
        __syncthreads();
    }
    if (tid < 4) out[tid] = red[0][tid];
}
}  // namespace

// d_X (2 n) and d_alpha (n) on the device; grad: 4 doubles on the host.  Synchronises the stream.
int launch_loglik_grad(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *d_X, const double *d_alpha, double *grad) {
    hipStream_t st = ctx->stream;
    const int64_t n = f->n;
    CovPlan pl;
    int rc = cov_plan(ctx, f, n, false, &pl);             // m = n: d_Bt (Np x Np) <- L^-T, d_C (Np x Np) <- -K^-1
    if (rc) return rc;
    const int64_t nrb = (n + 63) / 64;
    const int64_t nparts = nrb * pl.nP;
    rc = tgp_ensure_scratch2(ctx, (size_t)(nparts * 4 + 4) * sizeof(double) > (size_t)pl.Mp * 1024 * sizeof(double)
                                      ? (size_t)(nparts * 4 + 4) * sizeof(double) : (size_t)pl.Mp * 1024 * sizeof(double));
    if (rc) return rc;
    TGP_HIP(hipMemsetAsync(pl.d_Bt, 0, (size_t)pl.Mp * pl.Np * 8, st));
    TGP_HIP(hipMemsetAsync(pl.d_C, 0, (size_t)pl.Mp * pl.Mp * 8, st));
    ident_panels_kernel<<<(unsigned)(pl.Np / 256), 256, 0, st>>>(pl.d_Bt, pl.Mp, pl.Np);
    rc = cov_substitute(ctx, f, pl, true);
    if (rc) return rc;
    const int64_t mt = pl.Mp / TGP_TB;
    kinv_syrk_kernel<<<(unsigned)(mt * (mt + 1) / 2), 256, 0, st>>>(pl.d_C, pl.d_Bt, pl.Mp, pl.nP);
    double *partial = (double *)ctx->scratch2;             // the substitution's staging buffer is free again
    loglik_grad_kernel<<<dim3((unsigned)pl.nP, (unsigned)nrb), 256, 0, st>>>(make_kparams(k), d_X, d_alpha, pl.d_C, pl.Mp, n, partial);
    loglik_grad_reduce_kernel<<<1, 256, 0, st>>>(partial, nparts, partial + nparts * 4);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipEventRecord(ctx->ev[4], st));
    TGP_HIP(hipMemcpyAsync(grad, partial + nparts * 4, 4 * sizeof(double), hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    return 0;
}

static int loglik_grad_kind_check(tgp_ctx *ctx, const tgp_kernel *k) {
    if (kind_to_ke(k->kind) == KE_GAUSS) return 0;
    ctx->err = "tgp_gp_loglik_grad: analytic derivatives exist for the Gaussian kernels only (RBF, AnisotropicRBF), as in the "
               "reference (treegp/kernels.py:128-150)";
    return -1;
}

extern "C" int tgp_gp_loglik_grad(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *X, int64_t n, const double *alpha,
                                  double *grad) {
    TGP_ARG(f && k && X && alpha && grad && n == f->n);
    if (loglik_grad_kind_check(ctx, k)) return -1;
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = tgp_ensure_io(ctx, (size_t)3 * n * sizeof(double));          // coordinates and alpha: outside the scratch areas
    if (rc) return rc;
    double *d_X = (double *)tgp_io_buffer(ctx), *d_alpha = d_X + 2 * n;
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(d_X, X, 2 * n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_alpha, alpha, n * 8, hipMemcpyHostToDevice, st));
    rc = launch_loglik_grad(ctx, f, k, d_X, d_alpha, grad);
    if (rc) return rc;
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[4]));
    ctx->timings[3] = ms;
    return 0;
}

// K build + factorisation + solve + gradient for data that live on the device (tgp_d_gp_solve followed by the above, the
// factor going back to the context's cache instead of through a handle): one call per evaluation of a gradient-driven fit.
extern "C" int tgp_d_gp_solve_grad(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_y,
                                   const double *d_yerr, double *logdet, double *ydota, double *grad) {
    TGP_ARG(k && d_X && d_y && grad && n > 0);
    if (loglik_grad_kind_check(ctx, k)) return -1;
    TGP_HIP(hipSetDevice(ctx->device));
    int rc = tgp_ensure_io(ctx, (size_t)n * sizeof(double));
    if (rc) return rc;
    double *d_alpha = (double *)tgp_io_buffer(ctx);
    tgp_factor *f = nullptr;
    rc = tgp_d_gp_solve(ctx, k, d_X, n, d_y, d_yerr, d_alpha, logdet, ydota, &f);
    if (rc) return rc;                                      // > 0: not positive definite, nothing was kept
    rc = launch_loglik_grad(ctx, f, k, d_X, d_alpha, grad);
    float ms = 0.f;
    if (!rc && hipEventElapsedTime(&ms, ctx->ev[3], ctx->ev[4]) == hipSuccess) ctx->timings[3] = ms;
    tgp_factor_release_to_cache(ctx, f);
    return rc;
}
