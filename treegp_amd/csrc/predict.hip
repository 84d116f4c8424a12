// Fused GP mean prediction (seam S3): ys[q] = sum_i amp k(Xs_q, X_i) alpha_i, i.e.
// kernel(X2, Y=X1) followed by HT @ alpha (treegp/gp_interp.py:177,183) without ever
// materialising the (M, N) cross-kernel matrix.  fp64-VALU-bound (one transcendental per pair);
// HBM traffic is just the coordinates.
//
// One thread per query point; the training points (x, y, alpha) are staged through LDS in
// tiles of 256 and read back as wave-wide broadcasts.  The training set is split over
// gridDim.y so that small M still fills the chip; partial sums are combined in a fixed order
// by a second kernel (bitwise reproducible, no atomics).
#include "tgp_internal.h"
#include "kernel_eval.h"

namespace {
constexpr int PT = 256;   // training points per LDS tile

template <int KE>
__global__ __launch_bounds__(256) void predict_partial_kernel(KParams p, const double *__restrict__ X, int64_t n,
                                                              const double *__restrict__ alpha,
                                                              const double *__restrict__ Xs, int64_t m,
                                                              double *__restrict__ partial, int64_t chunk) {
    __shared__ double sx[PT], sy[PT], sa[PT];
    __shared__ double vk_tab[KE == KE_GAUSS ? 1 : 6 * K56_NDEG];       // von Karman: the Chebyshev table, gathered per lane
    if constexpr (KE != KE_GAUSS) vonkarman_stage_table(vk_tab);       // (the first barrier of the loop below publishes it)
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk;
    const int64_t i_end = (i_begin + chunk < n) ? i_begin + chunk : n;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = Xs[2 * q]; yq = Xs[2 * q + 1]; }
    double acc = 0.0;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += PT) {
        const int64_t i = i0 + tid;
        __syncthreads();
        if (i < i_end) { sx[tid] = X[2 * i]; sy[tid] = X[2 * i + 1]; sa[tid] = alpha[i]; }
        else { sx[tid] = 0.0; sy[tid] = 0.0; sa[tid] = 0.0; }
        __syncthreads();
        const int cnt = (int)((i_end - i0 < PT) ? (i_end - i0) : PT);
        if (cnt == PT) {
#pragma unroll 4
            for (int t = 0; t < PT; ++t) acc += kernel_value_tab<KE>(p, xq - sx[t], yq - sy[t], vk_tab) * sa[t];
        } else {
            for (int t = 0; t < cnt; ++t) acc += kernel_value_tab<KE>(p, xq - sx[t], yq - sy[t], vk_tab) * sa[t];
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc;
}

// ---- Gaussian fast path ---------------------------------------------------------------------------
// exp(-0.5 d^T invLam d) with invLam = L L^T is 2^-(|u|^2) for u = sqrt(0.5 log2 e) L^T d.  Coordinates
// are transformed once per call (O(n + m)), which leaves 2 sub + 1 mul + 1 fma for the exponent, and
// 2^-s needs no range checks for s >= 0: 22 fp64 instructions per pair instead of 32.  The rounding of
// the exponent differs from the reference's a dx^2 + 2 b dx dy + c dy^2 by ~1e-16 |q| relative
// (parity tests: 1e-10 on predicted values).  The amplitude is applied in the reduction.
__global__ __launch_bounds__(256) void predict_transform_kernel(const double *__restrict__ X, int64_t n, double t00,
                                                                double t10, double t11, double *__restrict__ U) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = X[2 * i], y = X[2 * i + 1];
    U[2 * i] = t00 * x + t10 * y;
    U[2 * i + 1] = t11 * y;
}

// (the K build's variant, tgp_exp2_neg in kernel_eval.h -- magic-constant rounding and an integer add into the exponent instead of
// rint / cvt / ldexp, plus a clamp -- is one instruction longer and measured 5 % slower here: 12.6 vs 12.0 ms per 1.7e10 pairs;
// v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 issue at the full fp64 rate on gfx950)
__device__ __forceinline__ double exp2_neg(double s) {        // 2^(-s), s >= 0
    const double t = -s;
    const double k = rint(t);
    const double f = t - k;                                    // exact, |f| <= 0.5
    double p = 1.3691488853904128881e-12;                      // Taylor of 2^f = sum (f ln2)^i / i!, i <= 13
    p = fma(p, f, 2.5678435993488205142e-11);
    p = fma(p, f, 4.4455382718708114976e-10);
    p = fma(p, f, 7.0549116208011233299e-9);
    p = fma(p, f, 1.0178086009239699727e-7);
    p = fma(p, f, 1.3215486790144309488e-6);
    p = fma(p, f, 1.525273380405984028e-5);
    p = fma(p, f, 1.5403530393381609954e-4);
    p = fma(p, f, 1.3333558146428443423e-3);
    p = fma(p, f, 9.618129107628477162e-3);
    p = fma(p, f, 5.5504108664821579953e-2);
    p = fma(p, f, 2.4022650695910071233e-1);
    p = fma(p, f, 6.9314718055994530942e-1);
    p = fma(p, f, 1.0);
    return ldexp(p, (int)k);
}

// Where its rate comes from: the inner loop is 22 fp64 VALU instructions per pair (13 fma of the polynomial, 3 add, 2 fmac, rndne,
// cvt, ldexp, mul) and nothing else but one ds_read_b128 per 1.3 pairs, i.e. 256 CUs x 4 SIMDs x 16 lanes x f / 22 pairs/s:
// 1.79e12 at 2.4 GHz, 1.64e12 at the 2.2 GHz (1.25 kW) the chip settles at under this kernel; measured back to back 1.50e12
// (tools/predict_clock.py), 1.38 - 1.43e12 as one launch of a bench step.  Two
// queries per thread (half the LDS reads per pair) changed nothing (3.10 vs 3.14 ms at N = 32 768, M = 131 072): issue-bound.
__global__ __launch_bounds__(256) void predict_gauss_fast_kernel(const double *__restrict__ U, int64_t n,
                                                                 const double *__restrict__ alpha,
                                                                 const double *__restrict__ Us, int64_t m,
                                                                 double *__restrict__ partial, int64_t chunk) {
    __shared__ double sx[PT], sy[PT], sa[PT];
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk;
    const int64_t i_end = (i_begin + chunk < n) ? i_begin + chunk : n;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = Us[2 * q]; yq = Us[2 * q + 1]; }
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += PT) {
        const int64_t i = i0 + tid;
        __syncthreads();
        if (i < i_end) { sx[tid] = U[2 * i]; sy[tid] = U[2 * i + 1]; sa[tid] = alpha[i]; }
        else { sx[tid] = 0.0; sy[tid] = 0.0; sa[tid] = 0.0; }          // alpha = 0: padded entries add nothing
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < PT; t += 2) {
            const double dx0 = xq - sx[t], dy0 = yq - sy[t];
            const double dx1 = xq - sx[t + 1], dy1 = yq - sy[t + 1];
            acc0 = fma(exp2_neg(fma(dx0, dx0, dy0 * dy0)), sa[t], acc0);
            acc1 = fma(exp2_neg(fma(dx1, dx1, dy1 * dy1)), sa[t + 1], acc1);
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc0 + acc1;
}

__global__ __launch_bounds__(256) void predict_reduce_kernel(const double *__restrict__ partial, int64_t m, int nsplit,
                                                             double scale, double *__restrict__ ys) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * m + q];
    ys[q] = scale * s;
}
}  // namespace

int launch_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_alpha,
                   const double *d_Xs, int64_t m, double *d_ys) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0);
    TGP_ARG(n > 0 && m > 0);
    const KParams p = make_kparams(k);
    const int64_t qblocks = (m + 255) / 256;
    // enough workgroups for ~8 per CU, but never split finer than one LDS tile
    int64_t nsplit = (2048 + qblocks - 1) / qblocks;
    const int64_t maxsplit = (n + PT - 1) / PT;
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    int64_t chunk = (n + nsplit - 1) / nsplit;
    chunk = (chunk + PT - 1) / PT * PT;
    nsplit = (n + chunk - 1) / chunk;
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    int rc = tgp_ensure_scratch(ctx, rup((size_t)nsplit * m * 8) + rup(2 * n * 8) + rup(2 * m * 8));
    if (rc) return rc;
    double *partial = (double *)ctx->scratch;
    dim3 grid((unsigned)qblocks, (unsigned)nsplit), block(256);
    static const bool no_fast = getenv("TGP_PREDICT_GENERIC") != nullptr;
    // invLam = L L^T (2x2 Cholesky); 1-D kernels have c = b = 0, i.e. l11 = 0
    const double l00 = (k->a > 0.0) ? sqrt(k->a) : 0.0;
    const double l10 = (l00 > 0.0) ? k->b / l00 : 0.0;
    const double d11 = k->c - l10 * l10;
    if (ke == KE_GAUSS && !no_fast && l00 > 0.0 && d11 >= 0.0) {
        const double sc = 0.84932180028801904272;            // sqrt(0.5 log2 e)
        const double t00 = sc * l00, t10 = sc * l10, t11 = sc * sqrt(d11);
        double *U = (double *)((char *)ctx->scratch + rup((size_t)nsplit * m * 8));
        double *Us = (double *)((char *)U + rup(2 * n * 8));
        predict_transform_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_X, n, t00, t10, t11, U);
        predict_transform_kernel<<<(unsigned)((m + 255) / 256), 256, 0, ctx->stream>>>(d_Xs, m, t00, t10, t11, Us);
        predict_gauss_fast_kernel<<<grid, block, 0, ctx->stream>>>(U, n, d_alpha, Us, m, partial, chunk);
        predict_reduce_kernel<<<(unsigned)qblocks, 256, 0, ctx->stream>>>(partial, m, (int)nsplit, k->amp, d_ys);
        TGP_HIP(hipGetLastError());
        return 0;
    }
    switch (ke) {
        case KE_GAUSS: predict_partial_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        case KE_VK: predict_partial_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        default: predict_partial_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
    }
    predict_reduce_kernel<<<(unsigned)qblocks, 256, 0, ctx->stream>>>(partial, m, (int)nsplit, 1.0, d_ys);
    TGP_HIP(hipGetLastError());
    return 0;
}
