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
            for (int t = 0; t < PT; ++t) acc += kernel_value<KE>(p, xq - sx[t], yq - sy[t]) * sa[t];
        } else {
            for (int t = 0; t < cnt; ++t) acc += kernel_value<KE>(p, xq - sx[t], yq - sy[t]) * sa[t];
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc;
}

__global__ __launch_bounds__(256) void predict_reduce_kernel(const double *__restrict__ partial, int64_t m, int nsplit,
                                                             double *__restrict__ ys) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * m + q];
    ys[q] = s;
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
    int rc = tgp_ensure_scratch(ctx, (size_t)nsplit * m * sizeof(double));
    if (rc) return rc;
    double *partial = (double *)ctx->scratch;
    dim3 grid((unsigned)qblocks, (unsigned)nsplit), block(256);
    switch (ke) {
        case KE_GAUSS: predict_partial_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        case KE_VK: predict_partial_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        default: predict_partial_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
    }
    predict_reduce_kernel<<<(unsigned)qblocks, 256, 0, ctx->stream>>>(partial, m, (int)nsplit, d_ys);
    TGP_HIP(hipGetLastError());
    return 0;
}
