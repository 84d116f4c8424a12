// Bootstrap of the TwoD pair correlation (treegp/two_pcf.py:342-362) with the geometry done ONCE.
//
// All n_boot resamples are multisets of the same base points: a pair of distinct base points (i, j) drawn c_i and
// c_j times in resample b contributes c_i c_j times, copies of one point have r = 0 and are excluded.  Which pixel a
// pair falls into does not depend on the resample, so
//   phase A  one traversal of the base catalogue's pairs (Morton-sorted 256-point tiles, bounding-box culling, the
//            exact binning rules of kk.hip) builds, per pixel, the list of pairs that deposit into it (both
//            orientations of a pair, as TreeCorr does);
//   phase B  every pixel's list is reduced with one thread per RESAMPLE: multiplicities come from a byte matrix
//            C[point][resample] (coalesced over resamples), the pair itself and its weights are wave-uniform.
//            Per pixel and resample three moments are kept, S0 = sum cc ww, S1 = sum cc ww (y_i + y_j),
//            S2 = sum cc ww y_i y_j, because the values are centred on the resample's own mean mu_b:
//            sum cc ww (y_i - mu)(y_j - mu) = S2 - mu S1 + mu^2 S0;
//   phase C  xi[b][pixel] = (S2 - mu_b S1 + mu_b^2 S0) / S0   (0 where S0 == 0).
// ~1e8 pair visits x n_boot/64 waves instead of n_boot traversals with ~1e10 weighted LDS deposits.
#include "tgp_internal.h"
#include <algorithm>
#include <thread>

void tgp_morton_keys(const double *x, const double *y, int64_t n, std::vector<uint32_t> &key, int &nbuckets);
void tgp_counting_sort_row(const int64_t *src, int64_t n, const std::vector<uint32_t> &key, int nbuckets,
                           std::vector<int64_t> &count, int64_t *dst);

namespace {
constexpr int KT = 256;
constexpr int MAXB2 = 32 * 32;

struct PairArgs {
    const double *x, *y;          // Morton-ordered base points
    const double *bbox;           // (ntile, 4) xmin, xmax, ymin, ymax per 256-point tile
    int64_t n;
    double min_sep, max_sep, bs, inv_bs, minsq;
    int nbins;
};

// int(x / bs) exactly as the IEEE division + truncation gives it (see kk.hip)
__device__ __forceinline__ int bin_of(double x, double bs, double inv_bs) {
    int i = (int)(x * inv_bs);
    const double r = fma(-(double)i, bs, x);
    const double eps = 1e-9 * bs;
    if (!(r > eps && r < bs - eps)) i = (int)(x / bs);
    return i;
}

// the two pixels a pair with separation (dx, dy) deposits into (-1: none); false if the pair is out of range
__device__ __forceinline__ bool pair_bins(const PairArgs &a, double dx, double dy, int &b0, int &b1) {
    const double rsq = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
    const double ad = fmax(fabs(dx), fabs(dy));
    if (!(rsq != 0.0 && rsq >= a.minsq && ad < a.max_sep)) return false;
    const int ix = bin_of(__dadd_rn(dx, a.max_sep), a.bs, a.inv_bs), iy = bin_of(__dadd_rn(dy, a.max_sep), a.bs, a.inv_bs);
    const int jx = bin_of(__dadd_rn(-dx, a.max_sep), a.bs, a.inv_bs), jy = bin_of(__dadd_rn(-dy, a.max_sep), a.bs, a.inv_bs);
    b0 = (ix >= 0 && ix < a.nbins && iy >= 0 && iy < a.nbins) ? iy * a.nbins + ix : -1;
    b1 = (jx >= 0 && jx < a.nbins && jy >= 0 && jy < a.nbins) ? jy * a.nbins + jx : -1;
    return true;
}

// Phase A.  FILL = false: per-pixel deposit counts.  FILL = true: the workgroup counts its own deposits, reserves
// a range in every pixel's list with one global atomic per pixel, then walks its pairs again and writes them.
template <bool FILL>
__global__ __launch_bounds__(256) void pair_list_kernel(PairArgs a, unsigned long long *__restrict__ counts,
                                                        const unsigned long long *__restrict__ offsets,
                                                        unsigned long long *__restrict__ cursor, int2 *__restrict__ entries) {
    __shared__ double sx[KT], sy[KT];
    __shared__ unsigned lcount[MAXB2];
    __shared__ unsigned long long lbase[FILL ? MAXB2 : 1];
    const int nb2 = a.nbins * a.nbins;
    const int tid = threadIdx.x;
    const int64_t ti = blockIdx.x;
    const int64_t ntile = (a.n + KT - 1) / KT;
    const double *bb = a.bbox;
    const double bxl = bb[ti * 4], bxh = bb[ti * 4 + 1], byl = bb[ti * 4 + 2], byh = bb[ti * 4 + 3];
    const int64_t i = ti * KT + tid;
    const bool ivalid = i < a.n;
    const double xi = ivalid ? a.x[i] : 0.0, yi = ivalid ? a.y[i] : 0.0;
    for (int pass = 0; pass < (FILL ? 2 : 1); ++pass) {
        for (int t = tid; t < nb2; t += 256) lcount[t] = 0;
        for (int64_t tj = ti + blockIdx.y; tj < ntile; tj += gridDim.y) {
            const double gx = fmax(0.0, fmax(bb[tj * 4] - bxh, bxl - bb[tj * 4 + 1]));
            const double gy = fmax(0.0, fmax(bb[tj * 4 + 2] - byh, byl - bb[tj * 4 + 3]));
            if (gx >= a.max_sep || gy >= a.max_sep) continue;                    // uniform
            __syncthreads();
            const int64_t j = tj * KT + tid;
            if (j < a.n) { sx[tid] = a.x[j]; sy[tid] = a.y[j]; }
            __syncthreads();
            const int cnt = (int)((a.n - tj * KT < KT) ? (a.n - tj * KT) : KT);
            const int t0 = (tj == ti) ? tid + 1 : 0;                              // unordered pairs: j > i
            if (!ivalid) continue;
            for (int t = t0; t < cnt; ++t) {
                int b0, b1;
                if (!pair_bins(a, sx[t] - xi, sy[t] - yi, b0, b1)) continue;
                if (!FILL || pass == 0) {
                    if (b0 >= 0) atomicAdd(&lcount[b0], 1u);
                    if (b1 >= 0) atomicAdd(&lcount[b1], 1u);
                } else {
                    const int2 e = make_int2((int)i, (int)(tj * KT + t));
                    if (b0 >= 0) entries[offsets[b0] + lbase[b0] + atomicAdd(&lcount[b0], 1u)] = e;
                    if (b1 >= 0) entries[offsets[b1] + lbase[b1] + atomicAdd(&lcount[b1], 1u)] = e;
                }
            }
        }
        __syncthreads();
        if (!FILL) {
            for (int t = tid; t < nb2; t += 256)
                if (lcount[t]) atomicAdd(&counts[t], (unsigned long long)lcount[t]);
        } else if (pass == 0) {
            for (int t = tid; t < nb2; t += 256) lbase[t] = lcount[t] ? atomicAdd(&cursor[t], (unsigned long long)lcount[t]) : 0ull;
        }
        __syncthreads();
    }
}

__global__ void scan_counts_kernel(const unsigned long long *__restrict__ counts, int nb2, unsigned long long *__restrict__ offsets) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long run = 0;
        for (int b = 0; b < nb2; ++b) { offsets[b] = run; run += counts[b]; }
        offsets[nb2] = run;
    }
}

struct PointW {
    double w, v;                  // weight and value (minus the base catalogue's mean) of a Morton-ordered base point
};

// Phase B: blockIdx.x = pixel, blockIdx.y = slice of its pair list; thread = resample
__global__ __launch_bounds__(1024) void moments_kernel(const int2 *__restrict__ entries, const unsigned long long *__restrict__ offsets,
                                                       const PointW *__restrict__ pts, const uint8_t *__restrict__ C, int nbp,
                                                       int n_boot, int nb2, double *__restrict__ S) {
    const int beta = blockIdx.x;
    const unsigned long long lo0 = offsets[beta], hi0 = offsets[beta + 1];
    const unsigned long long len = hi0 - lo0;
    const unsigned long long per = (len + gridDim.y - 1) / gridDim.y;
    unsigned long long lo = lo0 + per * blockIdx.y, hi = lo + per;
    if (hi > hi0) hi = hi0;
    if (lo >= hi) return;
    const int b = threadIdx.x;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll 4
    for (unsigned long long e = lo; e < hi; ++e) {
        const int2 ij = entries[e];                           // uniform over the workgroup
        const PointW pi = pts[ij.x], pj = pts[ij.y];
        const double cc = (double)((int)C[(int64_t)ij.x * nbp + b] * (int)C[(int64_t)ij.y * nbp + b]);
        const double t = cc * (pi.w * pj.w);
        s0 += t;
        s1 = fma(t, pi.v + pj.v, s1);
        s2 = fma(t, pi.v * pj.v, s2);
    }
    if (b < n_boot && s0 != 0.0) {
        double *o = S + ((int64_t)b * nb2 + beta) * 3;
        __hip_atomic_fetch_add(o, s0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(o + 1, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(o + 2, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// C[point][resample] from the host's resample-major rows (64 x 64 byte tiles through LDS)
__global__ __launch_bounds__(256) void transpose_mult_kernel(const uint8_t *__restrict__ Cb, int64_t n, int64_t n_boot, int nbp,
                                                             uint8_t *__restrict__ C) {
    __shared__ uint8_t tile[64][65];
    const int64_t i0 = (int64_t)blockIdx.x * 64, b0 = (int64_t)blockIdx.y * 64;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
    for (int r = r0; r < 64; r += 4) tile[r][c] = (b0 + r < n_boot && i0 + c < n) ? Cb[(b0 + r) * n + i0 + c] : (uint8_t)0;
    __syncthreads();
    for (int r = r0; r < 64; r += 4)
        if (i0 + r < n) C[(i0 + r) * nbp + b0 + c] = tile[c][r];
}

__global__ void boot_finalize_kernel(const double *__restrict__ S, const double *__restrict__ mean, int nb2, int64_t n_boot,
                                     double *__restrict__ xi) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_boot * nb2) return;
    const int64_t b = t / nb2;
    const double mu = mean[b];
    const double s0 = S[t * 3], s1 = S[t * 3 + 1], s2 = S[t * 3 + 2];
    xi[t] = (s0 != 0.0) ? (s2 - mu * s1 + mu * mu * s0) / s0 : 0.0;
}
}  // namespace

// returns 1 when the problem is outside what this formulation handles (the caller then uses the per-resample path)
int kk_bootstrap_lists(tgp_ctx *ctx, const double *x, const double *y, const double *v, const double *yerr_host, int64_t n,
                       const int64_t *idx, int64_t n_boot, double min_sep, double max_sep, int nbins, double *xi_out) {
    const int nb2 = nbins * nbins;
    const int nbp = (int)((n_boot + 63) / 64 * 64);
    if (n_boot < 8 || nbp > 1024 || nb2 > MAXB2 || n >= ((int64_t)1 << 31) || n < 2) return 1;
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t ntile = (n + KT - 1) / KT;

    // ---- host: Morton order of the base catalogue, multiplicity matrix, resample means ----------------------------
    std::vector<uint32_t> key;
    int nbuckets = 1;
    tgp_morton_keys(x, y, n, key, nbuckets);
    std::vector<int64_t> order0(n), count0;
    tgp_counting_sort_row(nullptr, n, key, nbuckets, count0, order0.data());
    std::vector<int32_t> rank(n);
    std::vector<double> xs(n), ys(n), bbox((size_t)ntile * 4);
    std::vector<PointW> pts(n);
    bool have_w = false;
    if (yerr_host) {
        double s = 0.0;                               // two_pcf.py:291-294: w = None if sum(y_err) == 0
        for (int64_t i = 0; i < n; ++i) s += yerr_host[i];
        have_w = s != 0.0;
    }
    // values are kept relative to the base catalogue's mean, so that the resample's own mean enters only as the small
    // shift mu_b - mu_0 and the moment formula does not cancel
    double mean0 = 0.0;
    for (int64_t i = 0; i < n; ++i) mean0 += v[i];
    mean0 /= (double)n;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t p = order0[t];
        rank[p] = (int32_t)t;
        xs[t] = x[p]; ys[t] = y[p];
        pts[t].v = v[p] - mean0;
        pts[t].w = have_w ? 1.0 / (yerr_host[p] * yerr_host[p]) : 1.0;
    }
    for (int64_t tl = 0; tl < ntile; ++tl) {
        double xl = xs[tl * KT], xh = xl, yl = ys[tl * KT], yh = yl;
        const int64_t end = std::min<int64_t>(n, (tl + 1) * KT);
        for (int64_t t = tl * KT; t < end; ++t) {
            xl = std::min(xl, xs[t]); xh = std::max(xh, xs[t]);
            yl = std::min(yl, ys[t]); yh = std::max(yh, ys[t]);
        }
        bbox[tl * 4] = xl; bbox[tl * 4 + 1] = xh; bbox[tl * 4 + 2] = yl; bbox[tl * 4 + 3] = yh;
    }
    // resample-major here (each host thread owns whole rows); transposed on the device.  Built in the context's pinned scratch:
    // 14.5 MB at 444 resamples of 32 768 points, which the runtime would otherwise pin for the upload and unpin behind the call
    uint8_t *Cm = nullptr;
    {
        void *pin = nullptr;
        int rcp = tgp_ensure_pinned(ctx, (size_t)n_boot * n, &pin);
        if (rcp) return rcp;
        Cm = (uint8_t *)pin;
        memset(Cm, 0, (size_t)n_boot * n);
    }
    std::vector<double> mean(n_boot, 0.0);
    std::vector<int> overflow(n_boot, 0);
    {
        const int nthr = (int)std::min<int64_t>(n_boot, std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
        auto work = [&](int tno) {
            for (int64_t b = tno; b < n_boot; b += nthr) {
                const int64_t *row = idx + b * n;
                uint8_t *col = Cm + (size_t)b * n;
                double s = 0.0;
                for (int64_t t = 0; t < n; ++t) {
                    const int64_t p = row[t];
                    uint8_t &c = col[rank[p]];
                    if (c == 255) overflow[b] = 1; else ++c;
                    s += v[p];
                }
                mean[b] = s / (double)n - mean0;
            }
        };
        if (nthr <= 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (int tno = 0; tno < nthr; ++tno) pool.emplace_back(work, tno);
            for (auto &th : pool) th.join();
        }
    }
    for (int64_t b = 0; b < n_boot; ++b)
        if (overflow[b]) return 1;                    // a point drawn more than 255 times: leave it to the general path

    // ---- device ----------------------------------------------------------------------------------------------------
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t fixed = 2 * rup(n * 8) + rup((size_t)ntile * 32) + rup(n * sizeof(PointW)) + rup((size_t)n * nbp) + rup((size_t)n_boot * n) +
                         3 * rup((size_t)(nb2 + 1) * 8) + rup((size_t)n_boot * nb2 * 24) + rup(n_boot * 8) +
                         rup((size_t)n_boot * nb2 * 8);
    int rc = tgp_ensure_scratch(ctx, fixed);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return p; };
    double *d_x = (double *)take(n * 8), *d_y = (double *)take(n * 8), *d_bbox = (double *)take((size_t)ntile * 32);
    PointW *d_pts = (PointW *)take(n * sizeof(PointW));
    uint8_t *d_C = (uint8_t *)take((size_t)n * nbp), *d_Cb = (uint8_t *)take((size_t)n_boot * n);
    unsigned long long *d_counts = (unsigned long long *)take((size_t)(nb2 + 1) * 8),
                       *d_offsets = (unsigned long long *)take((size_t)(nb2 + 1) * 8),
                       *d_cursor = (unsigned long long *)take((size_t)(nb2 + 1) * 8);
    double *d_S = (double *)take((size_t)n_boot * nb2 * 24), *d_mean = (double *)take(n_boot * 8),
           *d_xi = (double *)take((size_t)n_boot * nb2 * 8);
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(d_x, xs.data(), n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_y, ys.data(), n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_bbox, bbox.data(), (size_t)ntile * 32, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_pts, pts.data(), n * sizeof(PointW), hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_Cb, Cm, (size_t)n_boot * n, hipMemcpyHostToDevice, st));
    transpose_mult_kernel<<<dim3((unsigned)((n + 63) / 64), (unsigned)(nbp / 64)), 256, 0, st>>>(d_Cb, n, n_boot, nbp, d_C);
    TGP_HIP(hipMemcpyAsync(d_mean, mean.data(), n_boot * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemsetAsync(d_counts, 0, (size_t)(nb2 + 1) * 8, st));
    TGP_HIP(hipMemsetAsync(d_cursor, 0, (size_t)(nb2 + 1) * 8, st));
    TGP_HIP(hipMemsetAsync(d_S, 0, (size_t)n_boot * nb2 * 24, st));

    PairArgs a;
    a.x = d_x; a.y = d_y; a.bbox = d_bbox; a.n = n;
    a.min_sep = min_sep; a.max_sep = max_sep; a.minsq = min_sep * min_sep;
    a.nbins = nbins; a.bs = 2.0 * max_sep / nbins; a.inv_bs = 1.0 / a.bs;
    int jch = (int)((4096 + ntile - 1) / ntile);
    if (jch < 4) jch = 4;
    if (jch > ntile) jch = (int)ntile;
    dim3 gridA((unsigned)ntile, (unsigned)jch);
    pair_list_kernel<false><<<gridA, 256, 0, st>>>(a, d_counts, nullptr, nullptr, nullptr);
    scan_counts_kernel<<<1, 64, 0, st>>>(d_counts, nb2, d_offsets);
    unsigned long long total = 0;
    TGP_HIP(hipMemcpyAsync(&total, d_offsets + nb2, 8, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    if (total == 0) {
        for (int64_t t = 0; t < n_boot * nb2; ++t) xi_out[t] = 0.0;
        return 0;
    }
    if (total > ((unsigned long long)1 << 31)) return 1;      // > 16 GB of pair list: the per-resample path handles it
    // the pair list lives beside the fixed buffers in the grow-only scratch
    rc = tgp_ensure_scratch2(ctx, (size_t)total * sizeof(int2));
    if (rc) return rc;
    int2 *d_entries = (int2 *)ctx->scratch2;
    pair_list_kernel<true><<<gridA, 256, 0, st>>>(a, d_counts, d_offsets, d_cursor, d_entries);
    const unsigned slices = (unsigned)std::max<unsigned long long>(1, std::min<unsigned long long>(64, total / nb2 / 2048 + 1));
    moments_kernel<<<dim3((unsigned)nb2, slices), nbp, 0, st>>>(d_entries, d_offsets, d_pts, d_C, nbp, (int)n_boot, nb2, d_S);
    boot_finalize_kernel<<<(unsigned)((n_boot * nb2 + 255) / 256), 256, 0, st>>>(d_S, d_mean, nb2, n_boot, d_xi);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpyAsync(xi_out, d_xi, (size_t)n_boot * nb2 * 8, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[4] = ms;
    if (ctx->scratch2_bytes > ((size_t)256 << 20)) {      // a pair list of this size is not worth holding on to
        TGP_HIP(hipFree(ctx->scratch2));
        ctx->scratch2 = nullptr;
        ctx->scratch2_bytes = 0;
    }
    return 0;
}
