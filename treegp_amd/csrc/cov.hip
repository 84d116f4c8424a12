// Posterior covariance (seam S3b): cov = k(Xs, Xs) - HT (K + D)^-1 HT^T  (treegp/gp_interp.py:184-192)
// from the factor kept by tgp_gp_solve.  With Bt = HT L^-T (M x N) the result is
// Kss - Bt Bt^T, which reuses the factor instead of factorising a second time (the reference's
// own comment at gp_interp.py:189) and keeps the subtraction symmetric.
//
// Bt is produced by a right-looking block substitution over the 128-column blocks of L, every
// step being the same NT MFMA tile as the Cholesky (fp64 v_mfma_f64_16x16x4_f64):
//   Bt[:, kb]  = Bt[:, kb] W_kb^T                      (W_kb = inverse of the diagonal block)
//   Bt[:, c]  -= Bt[:, kb] L[c, kb]^T    for c > kb
// and finally cov = Kss - Bt Bt^T with depth Np.
#include "tgp_internal.h"
#include "kernel_eval.h"

typedef double d4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int KB = 16;
constexpr int LS = KB + 2;

// general NT tile with run-time strides/depth: MODE 0: C = A B^T, MODE 1: C -= A B^T
template <int MODE>
__device__ __forceinline__ void gemm_tile_rt(const double *a_ptr, int64_t lda, const double *b_ptr, int64_t ldb,
                                             double *c_ptr, int64_t ldc, int kdepth) {
    __shared__ __attribute__((aligned(16))) double lds[2][2][128 * LS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wr = w >> 1, wc = w & 1, l15 = lane & 15, l4 = lane >> 4;
    const int srow = tid >> 3, kp = (tid & 7) * 2;
    const double *ga = a_ptr + srow * lda + kp;
    const double *gb = b_ptr + srow * ldb + kp;

    d4 acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};

    const int fa = (wr * 64 + l15) * LS + l4, fb = (wc * 64 + l15) * LS + l4;
    const int nchunk = kdepth / KB;
    for (int c = 0; c < nchunk; ++c) {
        const int k0 = c * KB;
        double2 ra[4], rb[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ra[s] = *reinterpret_cast<const double2 *>(ga + (int64_t)s * 32 * lda + k0);
            rb[s] = *reinterpret_cast<const double2 *>(gb + (int64_t)s * 32 * ldb + k0);
        }
        __syncthreads();                               // previous chunk's fragment reads are done
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            *reinterpret_cast<double2 *>(&lds[0][0][(srow + 32 * s) * LS + kp]) = ra[s];
            *reinterpret_cast<double2 *>(&lds[0][1][(srow + 32 * s) * LS + kp]) = rb[s];
        }
        __syncthreads();
        const double *As = lds[0][0];
        const double *Bs = lds[0][1];
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
    }
    double *cbase = c_ptr + (wr * 64 + l4) * ldc + wc * 64 + l15;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *p = cbase + (int64_t)(m * 16 + 4 * r) * ldc + n * 16;
                if constexpr (MODE == 1) *p = *p - acc[m][n][r];
                else *p = acc[m][n][r];
            }
}

// grid (row tiles, column tiles): A tile rows advance with blockIdx.x, B tile rows with blockIdx.y
template <int MODE>
__global__ __launch_bounds__(256, 2) void gemm_rt_kernel(const double *A, int64_t lda, const double *B, int64_t ldb,
                                                         int64_t b_tile_stride, double *C, int64_t ldc,
                                                         int64_t c_col_stride, int kdepth) {
    const int64_t ti = blockIdx.x, tj = blockIdx.y;
    gemm_tile_rt<MODE>(A + ti * 128 * lda, lda, B + tj * b_tile_stride, ldb, C + ti * 128 * ldc + tj * c_col_stride,
                       ldc, kdepth);
}

// out (Mp, ld) = amp k(Xs_i, X_j) for i < m, j < n, zero elsewhere
template <int KE>
__global__ __launch_bounds__(256) void cross_padded_kernel(KParams p, const double *__restrict__ Xs, int64_t m,
                                                           const double *__restrict__ X, int64_t n, int self,
                                                           double *__restrict__ out, int64_t ld, int64_t ncols) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = blockIdx.y;
    if (j >= ncols) return;
    double v = 0.0;
    if (i < m && j < n) {
        v = kernel_value<KE>(p, Xs[2 * i] - X[2 * j], Xs[2 * i + 1] - X[2 * j + 1]);
        if (self && i == j) v = p.amp;
    }
    out[i * ld + j] = v;
}

int launch_cross_padded(tgp_ctx *ctx, const tgp_kernel *k, const double *d_Xs, int64_t m, const double *d_X, int64_t n,
                        int self, double *d_out, int64_t rows, int64_t ld, int64_t ncols) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0 && rows <= 65535);
    const KParams p = make_kparams(k);
    dim3 grid((unsigned)((ncols + 255) / 256), (unsigned)rows), block(256);
    switch (ke) {
        case KE_GAUSS: cross_padded_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, ld, ncols); break;
        case KE_VK: cross_padded_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, ld, ncols); break;
        default: cross_padded_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_Xs, m, d_X, n, self, d_out, ld, ncols); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" int tgp_gp_predict_cov(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *X, int64_t n,
                                  const double *Xs, int64_t m, double *cov) {
    TGP_ARG(f && k && X && Xs && cov && m > 0 && n == f->n);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t Np = f->Np;
    const int64_t Mp = (m + TGP_TB - 1) / TGP_TB * TGP_TB;
    TGP_ARG(Mp <= 65535);
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = rup(2 * n * 8) + rup(2 * m * 8) + rup((size_t)Mp * Np * 8) + rup((size_t)Mp * Mp * 8);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return (double *)p; };
    double *d_X = take(2 * n * 8), *d_Xs = take(2 * m * 8), *d_Bt = take((size_t)Mp * Np * 8),
           *d_C = take((size_t)Mp * Mp * 8);
    TGP_HIP(hipMemcpyAsync(d_X, X, 2 * n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_Xs, Xs, 2 * m * 8, hipMemcpyHostToDevice, st));
    // HT (gp_interp.py:177) and k(X2) (gp_interp.py:191), zero padded
    rc = launch_cross_padded(ctx, k, d_Xs, m, d_X, n, 0, d_Bt, Mp, Np, Np);
    if (rc) return rc;
    rc = launch_cross_padded(ctx, k, d_Xs, m, d_Xs, m, 1, d_C, Mp, Mp, Mp);
    if (rc) return rc;
    const int nb = (int)(Np / TGP_TB);
    const unsigned mt = (unsigned)(Mp / TGP_TB);
    for (int kb = 0; kb < nb; ++kb) {
        double *Bk = d_Bt + (int64_t)kb * TGP_TB;
        gemm_rt_kernel<0><<<dim3(mt, 1), 256, 0, st>>>(Bk, Np, f->d_W + (int64_t)kb * TGP_TB * TGP_TB, TGP_TB, 0, Bk, Np, 0,
                                                      TGP_TB);
        const int nc = nb - kb - 1;
        if (nc > 0) {
            const int64_t p = kb >> 1;
            const double *Lcol = f->d_A + panel_off(p, Np) + ((int64_t)(kb + 1) * TGP_TB - p * TGP_PW) * TGP_PW +
                                 (kb & 1) * TGP_TB;
            gemm_rt_kernel<1><<<dim3(mt, (unsigned)nc), 256, 0, st>>>(Bk, Np, Lcol, TGP_PW, (int64_t)TGP_TB * TGP_PW,
                                                                      Bk + TGP_TB, Np, TGP_TB, TGP_TB);
        }
    }
    gemm_rt_kernel<1><<<dim3(mt, mt), 256, 0, st>>>(d_Bt, Np, d_Bt, Np, (int64_t)TGP_TB * Np, d_C, Mp, TGP_TB, (int)Np);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpy2DAsync(cov, (size_t)m * 8, d_C, (size_t)Mp * 8, (size_t)m * 8, (size_t)m, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    return 0;
}
