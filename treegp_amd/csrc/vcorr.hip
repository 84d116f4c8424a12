// Vector (astrometric-residual) 2-point correlation, brute force over all pairs: the accumulation
// treegp/utils.py:5-74 (`vcorr`) does with np.triu_indices + np.histogram, as one tiled pair loop.
//
// For every pair i < j with separation d = p_j - p_i (complex d = ddx + i ddy), log|d| is binned
// on the uniform edges np.histogram builds for (bins, range): index from ((x - first) / (last -
// first)) * bins, then corrected against the edge array exactly as numpy does, last edge inclusive.
// Per bin: pair count, sum log|d|, sum (dx_i dx_j + dy_i dy_j)            -> xi_+
//          sum v_i v_j          (v = dx + i dy, complex)                  -> xi_z2
//          sum v_i v_j conj(d)^2 / |d|^2                                  -> xi_- + i xi_x
//
// Layout as kk.hip: thread = one i-point, j-points staged through LDS in tiles of 256 and read
// back as broadcasts, per-wave private LDS histograms (fp64 ds_add), one global atomic flush per
// workgroup.  VALU + LDS-atomic bound; HBM traffic is 32 n bytes.
#include "tgp_internal.h"

namespace {
constexpr int VT = 256;
constexpr int NACC = 7;
constexpr int MAXBINS = 512;

struct VArgs {
    const double *x, *y, *dx, *dy, *edges;
    int64_t n;
    double first, last, denom;
    int bins;
};

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__global__ __launch_bounds__(256) void vcorr_pairs_kernel(VArgs a, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nb = a.bins;
    double *sx = smem, *sy = smem + VT, *sdx = smem + 2 * VT, *sdy = smem + 3 * VT;
    double *edges = smem + 4 * VT;                 // nb + 1
    double *hist = edges + nb + 1;                 // [4 waves][NACC][nb]
    const int tid = threadIdx.x, wave = tid >> 6;
    double *myh = hist + (size_t)wave * NACC * nb;
    for (int t = tid; t < 4 * NACC * nb; t += 256) hist[t] = 0.0;
    for (int t = tid; t <= nb; t += 256) edges[t] = a.edges[t];

    const int64_t ti = blockIdx.x;
    const int64_t ntile = (a.n + VT - 1) / VT;
    const int64_t i = ti * VT + tid;
    const bool ivalid = i < a.n;
    double xi = 0, yi = 0, ui = 0, vi = 0;
    if (ivalid) { xi = a.x[i]; yi = a.y[i]; ui = a.dx[i]; vi = a.dy[i]; }
    for (int64_t tj = ti + blockIdx.y; tj < ntile; tj += gridDim.y) {
        __syncthreads();
        const int64_t j = tj * VT + tid;
        if (j < a.n) { sx[tid] = a.x[j]; sy[tid] = a.y[j]; sdx[tid] = a.dx[j]; sdy[tid] = a.dy[j]; }
        __syncthreads();
        const int cnt = (int)((a.n - tj * VT < VT) ? (a.n - tj * VT) : VT);
        const int t0 = (tj == ti) ? tid + 1 : 0;          // i < j
        if (!ivalid) continue;
        for (int t = t0; t < cnt; ++t) {
            const double ddx = sx[t] - xi, ddy = sy[t] - yi;
            const double r = hypot(ddx, ddy);             // np.absolute of the complex separation
            const double lr = log(r);
            if (!(lr >= a.first && lr <= a.last)) continue;
            int b = (int)(__dmul_rn(__ddiv_rn(__dsub_rn(lr, a.first), a.denom), (double)nb));
            if (b >= nb) b = nb - 1;
            if (b < 0) b = 0;
            if (lr < edges[b]) --b;                       // numpy's one-ulp corrections against the edges
            else if (lr >= edges[b + 1] && b != nb - 1) ++b;
            const double uj = sdx[t], vj = sdy[t];
            const double plus = __dadd_rn(__dmul_rn(ui, uj), __dmul_rn(vi, vj));
            const double zr = __dsub_rn(__dmul_rn(ui, uj), __dmul_rn(vi, vj));        // v_i v_j
            const double zi = __dadd_rn(__dmul_rn(ui, vj), __dmul_rn(vi, uj));
            // twice times conj(d) = ddx - i ddy, then over |d|^2
            const double m1r = __dadd_rn(__dmul_rn(zr, ddx), __dmul_rn(zi, ddy));
            const double m1i = __dsub_rn(__dmul_rn(zi, ddx), __dmul_rn(zr, ddy));
            const double m2r = __dadd_rn(__dmul_rn(m1r, ddx), __dmul_rn(m1i, ddy));
            const double m2i = __dsub_rn(__dmul_rn(m1i, ddx), __dmul_rn(m1r, ddy));
            const double rsq = __dadd_rn(__dmul_rn(ddx, ddx), __dmul_rn(ddy, ddy));
            lds_add(myh + b, 1.0);
            lds_add(myh + nb + b, lr);
            lds_add(myh + 2 * nb + b, plus);
            lds_add(myh + 3 * nb + b, zr);
            lds_add(myh + 4 * nb + b, zi);
            lds_add(myh + 5 * nb + b, __ddiv_rn(m2r, rsq));
            lds_add(myh + 6 * nb + b, __ddiv_rn(m2i, rsq));
        }
    }
    __syncthreads();
    for (int t = tid; t < NACC * nb; t += 256) {
        const double s = hist[t] + hist[NACC * nb + t] + hist[2 * NACC * nb + t] + hist[3 * NACC * nb + t];
        if (s != 0.0) __hip_atomic_fetch_add(out + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
}  // namespace

extern "C" int tgp_vcorr(tgp_ctx *ctx, const double *x, const double *y, const double *dx, const double *dy, int64_t n,
                         const double *edges, int nbins, double *acc_out) {
    TGP_ARG(x && y && dx && dy && edges && acc_out && n >= 1 && nbins >= 1 && nbins <= MAXBINS);
    for (int i = 1; i <= nbins; ++i) TGP_ARG(edges[i] > edges[i - 1]);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t accb = (size_t)NACC * nbins * 8;
    const size_t need = 4 * rup(n * 8) + rup((size_t)(nbins + 1) * 8) + rup(accb);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return (double *)p; };
    double *d_x = take(n * 8), *d_y = take(n * 8), *d_dx = take(n * 8), *d_dy = take(n * 8);
    double *d_edges = take((size_t)(nbins + 1) * 8), *d_acc = take(accb);
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_dx, dx, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_dy, dy, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_edges, edges, (size_t)(nbins + 1) * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemsetAsync(d_acc, 0, accb, st));
    VArgs a;
    a.x = d_x; a.y = d_y; a.dx = d_dx; a.dy = d_dy; a.edges = d_edges; a.n = n;
    a.first = edges[0]; a.last = edges[nbins]; a.denom = edges[nbins] - edges[0]; a.bins = nbins;
    const int64_t ntile = (n + VT - 1) / VT;
    int jch = (int)((8192 + ntile - 1) / ntile);
    if (jch < 4) jch = 4;
    if (jch > ntile) jch = (int)ntile;
    const size_t shm = (size_t)(4 * VT + nbins + 1 + 4 * NACC * nbins) * sizeof(double);
    TGP_HIP(hipFuncSetAttribute((const void *)vcorr_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    vcorr_pairs_kernel<<<dim3((unsigned)ntile, (unsigned)jch), 256, shm, st>>>(a, d_acc);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpyAsync(acc_out, d_acc, accb, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[4] = ms;
    return 0;
}
