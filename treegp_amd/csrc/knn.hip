// Mean function at arbitrary positions: uniform mean of the k nearest (Euclidean) neighbours in the
// "meanify" table -- what KNeighborsRegressor(n_neighbors=k).fit(X0, y0).predict(X) computes at
// treegp/gp_interp.py:236-238.  Brute force: one thread per query, the table streamed through LDS,
// the k best (distance^2, value) kept sorted in registers.  O(m n0) distance evaluations; the table
// has a few thousand rows, so even m = 10^6 queries is well under a millisecond of VALU work.
#include "tgp_internal.h"

namespace {
constexpr int KMAX = 16;
constexpr int KT = 256;

template <int K>
__global__ __launch_bounds__(256) void knn_mean_kernel(const double *__restrict__ X0, const double *__restrict__ y0,
                                                       int64_t n0, const double *__restrict__ X, int64_t m,
                                                       double *__restrict__ out) {
    __shared__ double sx[KT], sy[KT], sv[KT];
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = X[2 * q]; yq = X[2 * q + 1]; }
    double bd[K], bv[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { bd[j] = __builtin_huge_val(); bv[j] = 0.0; }
    for (int64_t i0 = 0; i0 < n0; i0 += KT) {
        __syncthreads();
        const int64_t i = i0 + tid;
        if (i < n0) { sx[tid] = X0[2 * i]; sy[tid] = X0[2 * i + 1]; sv[tid] = y0[i]; }
        __syncthreads();
        const int cnt = (int)((n0 - i0 < KT) ? (n0 - i0) : KT);
        for (int t = 0; t < cnt; ++t) {
            const double dx = sx[t] - xq, dy = sy[t] - yq;
            double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if (d < bd[K - 1]) {                     // strict: earlier table rows win ties
                double v = sv[t];
#pragma unroll
                for (int j = 0; j < K; ++j) {        // insertion into the sorted list
                    if (d < bd[j]) {
                        const double td = bd[j], tv = bv[j];
                        bd[j] = d; bv[j] = v;
                        d = td; v = tv;
                    }
                }
            }
        }
    }
    if (q < m) {
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < K; ++j) s += bv[j];      // nearest first, like np.mean over the sorted neighbours
        out[q] = s / (double)K;
    }
}

template <int K>
void run_knn(hipStream_t st, const double *X0, const double *y0, int64_t n0, const double *X, int64_t m, double *out) {
    knn_mean_kernel<K><<<(unsigned)((m + 255) / 256), 256, 0, st>>>(X0, y0, n0, X, m, out);
}
}  // namespace

extern "C" int tgp_knn_mean(tgp_ctx *ctx, const double *X0, const double *y0, int64_t n0, const double *X, int64_t m,
                            int k, double *out) {
    TGP_ARG(X0 && y0 && X && out && n0 >= k && m > 0 && k >= 1 && (k <= 8 || k == KMAX));
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    int rc = tgp_ensure_scratch(ctx, rup(2 * n0 * 8) + rup(n0 * 8) + rup(2 * m * 8) + rup(m * 8));
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return (double *)p; };
    double *d_X0 = take(2 * n0 * 8), *d_y0 = take(n0 * 8), *d_X = take(2 * m * 8), *d_o = take(m * 8);
    TGP_HIP(hipMemcpyAsync(d_X0, X0, 2 * n0 * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_y0, y0, n0 * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_X, X, 2 * m * 8, hipMemcpyHostToDevice, st));
    switch (k) {
        case 1: run_knn<1>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 2: run_knn<2>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 3: run_knn<3>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 4: run_knn<4>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 5: run_knn<5>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 6: run_knn<6>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 7: run_knn<7>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        case 8: run_knn<8>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
        default: run_knn<KMAX>(st, d_X0, d_y0, n0, d_X, m, d_o); break;
    }
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpyAsync(out, d_o, m * 8, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    return 0;
}
