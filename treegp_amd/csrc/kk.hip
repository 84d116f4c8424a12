// Binned scalar-scalar pair correlation (seam S4): the job treecorr.KKCorrelation.process does
// for treegp/two_pcf.py:297-305 (TwoD pixels, bin_slop=0) and :330-334 (log bins), and the
// bootstrap loop of two_pcf.py:342-362, as one tiled brute-force pair loop with exact binning.
//
// TreeCorr is a third-party dependency that is not vendored in the reference; the binning
// rules restated here are its published ones (see oracle/gp_oracle.py: kk_twod / kk_log):
//   TwoD: every unordered pair with r != 0, r >= min_sep, max(|dx|,|dy|) < max_sep goes to pixel
//         (int((dx+max_sep)/bs), int((dy+max_sep)/bs)) of d = p_j - p_i and to the pixel of -d.
//   Log : bin int((ln r - ln min_sep)/bs) for min_sep <= r < max_sep, each pair once.
//
// Layout: thread = one i-point, j-points staged through LDS in tiles of 256 and read back as
// broadcasts; every wave keeps a private histogram in LDS (fp64 ds_add), flushed once per
// workgroup with global fp64 atomics.  VALU + LDS-atomic bound; HBM traffic is ~32 N bytes.
#include "tgp_internal.h"
#include <algorithm>
#include <thread>

namespace {
constexpr int KT = 256;           // points per tile
constexpr int MAXB2 = 32 * 32;    // max TwoD pixels (nbins <= 32)
constexpr int MAXBL = 256;        // max log bins

struct KKArgs {
    const double *x, *y, *v, *w;  // v: value before mean subtraction
    const int64_t *idx;           // (n_boot, n) point order: spatially sorted (resample) indices
    const double *mean;           // (n_boot) mean to subtract from v (nullptr: 0)
    const int32_t *idx32;         // bootstrap: (n_boot, n) distinct points of every resample, Morton order (replaces idx)
    const uint8_t *mult;          // bootstrap: (n_boot, n) multiplicity of each of them (weight = mult x w)
    const int64_t *cnt;           // (n_boot) entries in use per resample (bootstrap: number of DISTINCT points), or nullptr
    const double *bbox;           // (n_boot, ntile, 4) xmin, xmax, ymin, ymax of every 256-point tile
    int64_t n;
    double min_sep, max_sep, bs, inv_bs, minsq, maxsq, lmin;
    int nbins, jchunks;
    int part, nparts;             // this launch owns the i-tiles ti = part, part + nparts, ...  (multi-GPU sharding)
};

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// int(x / bs) for x >= 0 exactly as the IEEE division + truncation gives it, without dividing unless
// x sits within 1e-9 bin widths of a bin edge: q = x * (1/bs) can be off by an ulp or two, which only
// matters at an edge; r = x - i*bs (one rounding, fma) tells how far from the edges x is.
__device__ __forceinline__ int bin_of(double x, double bs, double inv_bs) {
    int i = (int)(x * inv_bs);
    const double r = fma(-(double)i, bs, x);
    const double eps = 1e-9 * bs;
    if (!(r > eps && r < bs - eps)) i = (int)(x / bs);
    return i;
}

// x_i, y_i bounding boxes of the tiles (one workgroup per tile and resample)
__global__ __launch_bounds__(256) void kk_bbox_kernel(KKArgs a, double *__restrict__ bbox) {
    __shared__ double r[4][256];
    const int tid = threadIdx.x;
    const int64_t boot = blockIdx.y, tile = blockIdx.x;
    const int64_t ntile = (a.n + KT - 1) / KT;
    const int64_t i = tile * KT + tid;
    double xv = 0, yv = 0;
    const int64_t npts = a.cnt ? a.cnt[boot] : a.n;
    const bool ok = i < npts;
    if (ok) { const int64_t s = a.idx32 ? (int64_t)a.idx32[boot * a.n + i] : a.idx[boot * a.n + i]; xv = a.x[s]; yv = a.y[s]; }
    const double big = __builtin_huge_val();
    r[0][tid] = ok ? xv : big; r[1][tid] = ok ? xv : -big; r[2][tid] = ok ? yv : big; r[3][tid] = ok ? yv : -big;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            r[0][tid] = fmin(r[0][tid], r[0][tid + o]); r[1][tid] = fmax(r[1][tid], r[1][tid + o]);
            r[2][tid] = fmin(r[2][tid], r[2][tid + o]); r[3][tid] = fmax(r[3][tid], r[3][tid + o]);
        }
        __syncthreads();
    }
    if (tid < 4) bbox[(boot * ntile + tile) * 4 + tid] = r[tid][0];
}

// out layout: [boot][NACC][nb]   NACC = 3 (TwoD: wkk, w, n) or 5 (Log: wkk, w, wr, wlogr, n)
template <bool TWOD>
__global__ __launch_bounds__(256) void kk_pairs_kernel(KKArgs a, double *__restrict__ out) {
    constexpr int NACC = TWOD ? 3 : 5;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nb = TWOD ? a.nbins * a.nbins : a.nbins;
    double *sx = smem, *sy = smem + KT, *sk = smem + 2 * KT, *sw = smem + 3 * KT;
    double *hist = smem + 4 * KT;                 // [4 waves][NACC][nb]
    const int tid = threadIdx.x, wave = tid >> 6;
    double *myh = hist + (size_t)wave * NACC * nb;
    for (int t = tid; t < 4 * NACC * nb; t += 256) hist[t] = 0.0;
    constexpr int NTOP = 4;                       // Log bins kept in registers (see the pair loop)
    const int topbase = a.nbins - NTOP;           // may be negative for fewer than NTOP bins: those slots then stay empty
    double racc[TWOD ? 1 : NTOP][5];
    if constexpr (!TWOD) {
#pragma unroll
        for (int q = 0; q < NTOP; ++q)
#pragma unroll
            for (int c = 0; c < 5; ++c) racc[q][c] = 0.0;
    }

    const int64_t boot = blockIdx.z;
    const int64_t *idx = a.idx ? a.idx + boot * a.n : nullptr;
    const int32_t *idx32 = a.idx32 ? a.idx32 + boot * a.n : nullptr;
    const uint8_t *mult = a.mult ? a.mult + boot * a.n : nullptr;
    const double mean = a.mean ? a.mean[boot] : 0.0;
    const int64_t ti = (int64_t)blockIdx.x * a.nparts + a.part;
    const int64_t ntile0 = (a.n + KT - 1) / KT;                   // row stride of the bbox table
    const int64_t npts = a.cnt ? a.cnt[boot] : a.n;
    const int64_t ntile = (npts + KT - 1) / KT;
    if (ti >= ntile) return;
    const double *bb = a.bbox + boot * ntile0 * 4;
    const double bxl = bb[ti * 4], bxh = bb[ti * 4 + 1], byl = bb[ti * 4 + 2], byh = bb[ti * 4 + 3];
    const int64_t i = ti * KT + tid;
    double xi = 0, yi = 0, ki = 0, wi = 0;
    const bool ivalid = i < npts;
    if (ivalid) {
        const int64_t s = idx32 ? (int64_t)idx32[i] : idx[i];
        xi = a.x[s]; yi = a.y[s]; ki = a.v[s] - mean; wi = a.w ? a.w[s] : 1.0;
        if (mult) wi *= (double)mult[i];
    }
    // j tiles tj >= ti, dealt round-robin over gridDim.y chunks; tiles out of reach are skipped whole
    for (int64_t tj = ti + blockIdx.y; tj < ntile; tj += gridDim.y) {
        {
            const double gx = fmax(0.0, fmax(bb[tj * 4] - bxh, bxl - bb[tj * 4 + 1]));
            const double gy = fmax(0.0, fmax(bb[tj * 4 + 2] - byh, byl - bb[tj * 4 + 3]));
            if (TWOD ? (gx >= a.max_sep || gy >= a.max_sep) : (gx * gx + gy * gy >= a.maxsq)) continue;   // uniform
        }
        __syncthreads();
        const int64_t j = tj * KT + tid;
        if (j < npts) {
            const int64_t s = idx32 ? (int64_t)idx32[j] : idx[j];
            double wj = a.w ? a.w[s] : 1.0;
            if (mult) wj *= (double)mult[j];
            sx[tid] = a.x[s]; sy[tid] = a.y[s]; sk[tid] = a.v[s] - mean; sw[tid] = wj;
        }
        __syncthreads();
        const int cnt = (int)((npts - tj * KT < KT) ? (npts - tj * KT) : KT);
        const int t0 = (tj == ti) ? tid + 1 : 0;          // unordered pairs: j > i
        if (!ivalid) continue;
        for (int t = 0; t < cnt; ++t) {
            const double dx = sx[t] - xi, dy = sy[t] - yi;
            const double rsq = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if constexpr (TWOD) {
                const double ad = fmax(fabs(dx), fabs(dy));
                if (t >= t0 && rsq != 0.0 && rsq >= a.minsq && ad < a.max_sep) {
                    const double ww = wi * sw[t];
                    const double wkk = ww * (ki * sk[t]);
                    const int ix = bin_of(__dadd_rn(dx, a.max_sep), a.bs, a.inv_bs);
                    const int iy = bin_of(__dadd_rn(dy, a.max_sep), a.bs, a.inv_bs);
                    if (ix >= 0 && ix < a.nbins && iy >= 0 && iy < a.nbins) {
                        const int b = iy * a.nbins + ix;
                        lds_add(myh + b, wkk); lds_add(myh + nb + b, ww); lds_add(myh + 2 * nb + b, 1.0);
                    }
                    const int jx = bin_of(__dadd_rn(-dx, a.max_sep), a.bs, a.inv_bs);
                    const int jy = bin_of(__dadd_rn(-dy, a.max_sep), a.bs, a.inv_bs);
                    if (jx >= 0 && jx < a.nbins && jy >= 0 && jy < a.nbins) {
                        const int b = jy * a.nbins + jx;
                        lds_add(myh + b, wkk); lds_add(myh + nb + b, ww); lds_add(myh + 2 * nb + b, 1.0);
                    }
                }
            } else {
                if (t >= t0 && rsq >= a.minsq && rsq < a.maxsq) {
                    const double lr = 0.5 * log(rsq);
                    const int b = bin_of(lr - a.lmin, a.bs, a.inv_bs);      // = (int)((lr - lmin) / bs), dividing only next to an edge
                    if (b >= 0 && b < a.nbins) {
                        const double ww = wi * sw[t];
                        const double v0 = ww * (ki * sk[t]), v2 = ww * sqrt(rsq), v3 = ww * lr;
                        // Log bins of a 2-D field are crowded at the top (populations grow like r^2: the last four of 20 bins
                        // take ~85 % of the pairs) and same-address fp64 LDS atomics retire one lane per clock: the kernel sat
                        // exactly on that ceiling (tools/probes/lds_atomic_ceiling.hip).  The top NTOP bins are therefore
                        // accumulated in registers, one set per lane, with 0/1 weights (5 FMAs per bin and pair, no
                        // divergence), and only the other bins go through the LDS atomics.
                        const int k = b - topbase;
                        if (k >= 0) {
#pragma unroll
                            for (int q = 0; q < NTOP; ++q) {
                                const double m = (k == q) ? 1.0 : 0.0;
                                racc[q][0] = fma(m, v0, racc[q][0]);
                                racc[q][1] = fma(m, ww, racc[q][1]);
                                racc[q][2] = fma(m, v2, racc[q][2]);
                                racc[q][3] = fma(m, v3, racc[q][3]);
                                racc[q][4] += m;
                            }
                        } else {
                            lds_add(myh + b, v0);
                            lds_add(myh + nb + b, ww);
                            lds_add(myh + 2 * nb + b, v2);
                            lds_add(myh + 3 * nb + b, v3);
                            lds_add(myh + 4 * nb + b, 1.0);
                        }
                    }
                }
            }
        }
    }
    if constexpr (!TWOD) {
        // the lanes' register accumulators: fixed-order butterfly over the wave, lane 0 adds them to the wave's histogram
        // (nobody else touches it: the LDS atomics of this wave are complete in program order)
#pragma unroll
        for (int q = 0; q < NTOP; ++q)
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                double v = racc[q][c];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if ((tid & 63) == 0 && topbase + q >= 0 && v != 0.0) lds_add(myh + c * nb + topbase + q, v);
            }
    }
    __syncthreads();
    double *o = out + (size_t)boot * NACC * nb;
    for (int t = tid; t < NACC * nb; t += 256) {
        const double s = hist[t] + hist[NACC * nb + t] + hist[2 * NACC * nb + t] + hist[3 * NACC * nb + t];
        if (s != 0.0) glb_add(o + t, s);
    }
}

// mean[b] = mean(v[idx[b, :]])  (np.mean of the resampled values, two_pcf.py:297 `k=(y - np.mean(y))`)
__global__ __launch_bounds__(256) void boot_mean_kernel(const double *__restrict__ v, const int64_t *__restrict__ idx,
                                                        int64_t n, double *__restrict__ mean) {
    __shared__ double part[256];
    const int64_t b = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[idx[b * n + i]];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean[b] = part[0] / (double)n;
}

__global__ void inv_sq_kernel(const double *__restrict__ e, int64_t n, double *__restrict__ w) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) w[i] = 1.0 / (e[i] * e[i]);
}

// xi[b][bin] = wkk / w (0 where w == 0); optional copies of the other accumulators
__global__ void kk_finalize_kernel(const double *__restrict__ acc, int nacc, int nb, int64_t nboot,
                                   double *__restrict__ xi) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= nboot * nb) return;
    const int64_t b = t / nb;
    const int bin = (int)(t % nb);
    const double *a = acc + (size_t)b * nacc * nb;
    const double w = a[nb + bin];
    xi[t] = (w != 0.0) ? a[bin] / w : 0.0;
}

struct IoBuf {
    void *p = nullptr;
    size_t bytes = 0;
};
}  // namespace

// Z-order (Morton) key of every point on a 2^b x 2^b grid with ~32 points per cell, and the counting-sort
// permutation of `src` (indices into the points) by that key.  Spatially compact 256-point tiles let
// the pair kernel skip tile pairs that are farther apart than max_sep.
void tgp_morton_keys(const double *x, const double *y, int64_t n, std::vector<uint32_t> &key, int &nbuckets) {
    double xl = x[0], xh = x[0], yl = y[0], yh = y[0];
    for (int64_t i = 1; i < n; ++i) {
        xl = x[i] < xl ? x[i] : xl; xh = x[i] > xh ? x[i] : xh;
        yl = y[i] < yl ? y[i] : yl; yh = y[i] > yh ? y[i] : yh;
    }
    int b = 0;
    while (b < 8 && ((int64_t)1 << (2 * (b + 1))) * 32 <= n) ++b;
    const int nc = 1 << b;
    nbuckets = nc * nc;
    const double sxc = (xh > xl) ? nc / (xh - xl) : 0.0, syc = (yh > yl) ? nc / (yh - yl) : 0.0;
    key.resize(n);
    for (int64_t i = 0; i < n; ++i) {
        int cx = (int)((x[i] - xl) * sxc), cy = (int)((y[i] - yl) * syc);
        cx = cx < 0 ? 0 : (cx >= nc ? nc - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= nc ? nc - 1 : cy);
        uint32_t k = 0;
        for (int t = 0; t < b; ++t) k |= (uint32_t)((cx >> t) & 1) << (2 * t) | (uint32_t)((cy >> t) & 1) << (2 * t + 1);
        key[i] = k;
    }
}
void tgp_counting_sort_row(const int64_t *src, int64_t n, const std::vector<uint32_t> &key, int nbuckets,
                              std::vector<int64_t> &count, int64_t *dst) {
    count.assign(nbuckets + 1, 0);
    for (int64_t t = 0; t < n; ++t) ++count[key[src ? src[t] : t] + 1];
    for (int k = 0; k < nbuckets; ++k) count[k + 1] += count[k];
    for (int64_t t = 0; t < n; ++t) { const int64_t s = src ? src[t] : t; dst[count[key[s]]++] = s; }
}

static int kk_run(tgp_ctx *ctx, bool twod, const double *x, const double *y, const double *v, const double *w_host,
                  const double *yerr_host, int64_t n, const int64_t *idx, int64_t n_boot, double min_sep,
                  double max_sep, int nbins, std::vector<double> &acc_host, int part = 0, int nparts = 1) {
    TGP_ARG(x && y && v && n > 1 && nbins > 0 && n_boot >= 1);
    TGP_ARG(nparts >= 1 && part >= 0 && part < nparts);
    TGP_ARG(twod ? (nbins * nbins <= MAXB2) : (nbins <= MAXBL));
    TGP_ARG(max_sep > 0.0 && (twod || min_sep > 0.0));
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nacc = twod ? 3 : 5;
    const int nb = twod ? nbins * nbins : nbins;
    const size_t accb = (size_t)n_boot * nacc * nb * sizeof(double);
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const int64_t ntile0 = (n + KT - 1) / KT;
    const bool weighted_boot = twod && idx != nullptr;      // bootstrap: distinct points with multiplicity weights
    const size_t need = 4 * rup(n * 8) + rup((size_t)n_boot * n * 8) + rup(n_boot * 8) + rup(accb) +
                        rup((size_t)n_boot * ntile0 * 4 * 8) +
                        (weighted_boot ? rup((size_t)n_boot * n) + rup(n_boot * 8) : 0);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return p; };
    double *d_x = (double *)take(n * 8), *d_y = (double *)take(n * 8), *d_v = (double *)take(n * 8),
           *d_w = (double *)take(n * 8);
    int64_t *d_idx = (int64_t *)take((size_t)n_boot * n * 8);
    double *d_mean = (double *)take(n_boot * 8);
    double *d_acc = (double *)take(accb);
    double *d_bbox = (double *)take((size_t)n_boot * ntile0 * 4 * 8);
    uint8_t *d_mult = weighted_boot ? (uint8_t *)take((size_t)n_boot * n) : nullptr;
    int64_t *d_cnt = weighted_boot ? (int64_t *)take(n_boot * 8) : nullptr;
    // spatial order of every catalogue (the base points, or each bootstrap resample)
    std::vector<int64_t> sorted(weighted_boot ? 0 : (size_t)n_boot * n);
    std::vector<int32_t> sorted32(weighted_boot ? (size_t)n_boot * n : 0);     // bootstrap: 4-byte indices, 1-byte multiplicities
    std::vector<uint8_t> mult_host;
    std::vector<double> mean_host;
    std::vector<int64_t> cnt_host;
    if (weighted_boot) {
        // A resample is a multiset of the base points.  A pair of DISTINCT points (i, j) drawn c_i and c_j times
        // appears c_i c_j times in the resampled catalogue and pairs of copies of one point have r = 0 (excluded), so
        // the resample's sums are those of its distinct points with weights c w: ~63 % of the points, 40 % of the
        // pair tests.  Per resample: multiplicities, mean of the drawn values (two_pcf.py:297), and the distinct
        // points in the Morton order of the base catalogue.
        std::vector<uint32_t> key;
        int nbuckets = 1;
        tgp_morton_keys(x, y, n, key, nbuckets);
        std::vector<int64_t> order0(n), count0;
        tgp_counting_sort_row(nullptr, n, key, nbuckets, count0, order0.data());
        TGP_ARG(n < (int64_t)1 << 31);
        mult_host.assign((size_t)n_boot * n, 0);
        mean_host.assign(n_boot, 0.0);
        cnt_host.assign(n_boot, 0);
        const int nthr = (int)std::min<int64_t>(n_boot, std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
        auto work = [&](int tno) {
            std::vector<int32_t> mult(n);
            for (int64_t b = tno; b < n_boot; b += nthr) {
                std::fill(mult.begin(), mult.end(), 0);
                const int64_t *row = idx + b * n;
                double s = 0.0;
                for (int64_t t = 0; t < n; ++t) {
                    ++mult[row[t]];
                    s += v[row[t]];
                }
                mean_host[b] = s / (double)n;
                int64_t m = 0;
                for (int64_t t = 0; t < n; ++t) {
                    const int64_t p = order0[t];
                    int32_t c = mult[p];
                    while (c > 0) {                      // multiplicities above 255 (never seen in practice) take several entries
                        sorted32[b * n + m] = (int32_t)p;
                        mult_host[b * n + m] = (uint8_t)(c > 255 ? 255 : c);
                        c -= 255;
                        ++m;
                    }
                }
                cnt_host[b] = m;
            }
        };
        if (nthr <= 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (int tno = 0; tno < nthr; ++tno) pool.emplace_back(work, tno);
            for (auto &th : pool) th.join();
        }
    } else if (twod) {
        std::vector<uint32_t> key;
        int nbuckets = 1;
        tgp_morton_keys(x, y, n, key, nbuckets);
        // the resamples are independent: sort their index rows on a few host threads
        const int nthr = (int)std::min<int64_t>(n_boot, std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency())));
        auto work = [&](int tno) {
            std::vector<int64_t> count;
            for (int64_t b = tno; b < n_boot; b += nthr)
                tgp_counting_sort_row(idx ? idx + b * n : nullptr, n, key, nbuckets, count, sorted.data() + b * n);
        };
        if (nthr <= 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (int tno = 0; tno < nthr; ++tno) pool.emplace_back(work, tno);
            for (auto &th : pool) th.join();
        }
    } else {
        // log bins reach across most of the field: nothing to cull, and neighbouring i-points of a wave
        // would hit the same bin for the same j (LDS-atomic serialisation), so keep the caller's order
        for (int64_t b = 0; b < n_boot; ++b)
            for (int64_t t = 0; t < n; ++t) sorted[b * n + t] = idx ? idx[b * n + t] : t;
    }

    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(d_x, x, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_v, v, n * 8, hipMemcpyHostToDevice, st));
    bool have_w = false;
    if (w_host) {
        TGP_HIP(hipMemcpyAsync(d_w, w_host, n * 8, hipMemcpyHostToDevice, st));
        have_w = true;
    } else if (yerr_host) {
        double s = 0.0;                               // two_pcf.py:291-294: w = None if sum(y_err) == 0
        for (int64_t i = 0; i < n; ++i) s += yerr_host[i];
        if (s != 0.0) {
            TGP_HIP(hipMemcpyAsync(d_w, yerr_host, n * 8, hipMemcpyHostToDevice, st));
            inv_sq_kernel<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(d_w, n, d_w);
            have_w = true;
        }
    }
    if (weighted_boot) {
        TGP_HIP(hipMemcpyAsync(d_idx, sorted32.data(), (size_t)n_boot * n * 4, hipMemcpyHostToDevice, st));
        TGP_HIP(hipMemcpyAsync(d_mult, mult_host.data(), (size_t)n_boot * n, hipMemcpyHostToDevice, st));
        TGP_HIP(hipMemcpyAsync(d_cnt, cnt_host.data(), (size_t)n_boot * 8, hipMemcpyHostToDevice, st));
        TGP_HIP(hipMemcpyAsync(d_mean, mean_host.data(), (size_t)n_boot * 8, hipMemcpyHostToDevice, st));
    } else {
        TGP_HIP(hipMemcpyAsync(d_idx, sorted.data(), (size_t)n_boot * n * 8, hipMemcpyHostToDevice, st));
        if (idx) boot_mean_kernel<<<(unsigned)n_boot, 256, 0, st>>>(d_v, d_idx, n, d_mean);
    }
    TGP_HIP(hipMemsetAsync(d_acc, 0, accb, st));

    KKArgs a;
    a.x = d_x; a.y = d_y; a.v = d_v; a.w = have_w ? d_w : nullptr;
    a.idx = weighted_boot ? nullptr : d_idx;
    a.idx32 = weighted_boot ? (const int32_t *)d_idx : nullptr;
    a.mult = d_mult;
    a.mean = idx ? d_mean : nullptr;
    a.cnt = d_cnt;
    a.bbox = d_bbox;
    a.n = n;
    a.min_sep = min_sep; a.max_sep = max_sep;
    a.minsq = min_sep * min_sep; a.maxsq = max_sep * max_sep;
    a.nbins = nbins;
    a.part = part; a.nparts = nparts;
    if (twod) { a.bs = 2.0 * max_sep / nbins; a.lmin = 0.0; }
    else { a.bs = log(max_sep / min_sep) / nbins; a.lmin = log(min_sep); }
    a.inv_bs = 1.0 / a.bs;
    const int64_t ntile = (n + KT - 1) / KT;
    kk_bbox_kernel<<<dim3((unsigned)ntile, (unsigned)n_boot), 256, 0, st>>>(a, d_bbox);
    // enough workgroups to fill the chip and to even out the triangular j range of the i-tiles
    int jch = (int)((8192 + ntile * n_boot - 1) / (ntile * n_boot));
    if (jch < 4) jch = 4;
    if (jch > ntile) jch = (int)ntile;
    a.jchunks = jch;
    const size_t shm = (size_t)(4 * KT + 4 * nacc * nb) * sizeof(double);
    const int64_t my_tiles = (ntile - part + nparts - 1) / nparts;      // ti = part, part + nparts, ... < ntile
    dim3 grid((unsigned)my_tiles, (unsigned)jch, (unsigned)n_boot), block(256);
    if (my_tiles <= 0) {
    } else if (twod) {
        TGP_HIP(hipFuncSetAttribute((const void *)kk_pairs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        kk_pairs_kernel<true><<<grid, block, shm, st>>>(a, d_acc);
    } else {
        TGP_HIP(hipFuncSetAttribute((const void *)kk_pairs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        kk_pairs_kernel<false><<<grid, block, shm, st>>>(a, d_acc);
    }
    TGP_HIP(hipGetLastError());
    acc_host.resize((size_t)n_boot * nacc * nb);
    TGP_HIP(hipMemcpyAsync(acc_host.data(), d_acc, accb, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[4] = ms;
    return 0;
}

int kk_bootstrap_lists(tgp_ctx *ctx, const double *x, const double *y, const double *v, const double *yerr_host, int64_t n,
                       const int64_t *idx, int64_t n_boot, double min_sep, double max_sep, int nbins, double *xi_out);

extern "C" {

int tgp_kk_twod(tgp_ctx *ctx, const double *x, const double *y, const double *k, const double *w, int64_t n,
                double min_sep, double max_sep, int nbins, double *xi, double *weight, double *npairs) {
    std::vector<double> acc;
    int rc = kk_run(ctx, true, x, y, k, w, nullptr, n, nullptr, 1, min_sep, max_sep, nbins, acc);
    if (rc) return rc;
    const int nb = nbins * nbins;
    for (int b = 0; b < nb; ++b) {
        const double ww = acc[nb + b];
        if (xi) xi[b] = (ww != 0.0) ? acc[b] / ww : 0.0;
        if (weight) weight[b] = ww;
        if (npairs) npairs[b] = acc[2 * nb + b];
    }
    return 0;
}

int tgp_kk_log(tgp_ctx *ctx, const double *x, const double *y, const double *k, const double *w, int64_t n,
               double min_sep, double max_sep, int nbins, double *xi, double *weight, double *meanr,
               double *meanlogr, double *npairs) {
    std::vector<double> acc;
    int rc = kk_run(ctx, false, x, y, k, w, nullptr, n, nullptr, 1, min_sep, max_sep, nbins, acc);
    if (rc) return rc;
    const int nb = nbins;
    for (int b = 0; b < nb; ++b) {
        const double ww = acc[nb + b];
        const bool nz = ww != 0.0;
        if (xi) xi[b] = nz ? acc[b] / ww : 0.0;
        if (weight) weight[b] = ww;
        if (meanr) meanr[b] = nz ? acc[2 * nb + b] / ww : 0.0;
        if (meanlogr) meanlogr[b] = nz ? acc[3 * nb + b] / ww : 0.0;
        if (npairs) npairs[b] = acc[4 * nb + b];
    }
    return 0;
}

// raw accumulators of the i-tiles dealt to `part` of `nparts`: the sum over parts is the whole catalogue
// (multi-GPU: one part per rank, all-reduce, then xi = acc[0] / acc[1] on the host)
int tgp_kk_partial(tgp_ctx *ctx, int bin_type, const double *x, const double *y, const double *k, const double *w,
                   int64_t n, double min_sep, double max_sep, int nbins, int part, int nparts, double *acc_out) {
    TGP_ARG(acc_out && (bin_type == 0 || bin_type == 1));
    std::vector<double> acc;
    int rc = kk_run(ctx, bin_type == 0, x, y, k, w, nullptr, n, nullptr, 1, min_sep, max_sep, nbins, acc, part, nparts);
    if (rc) return rc;
    memcpy(acc_out, acc.data(), acc.size() * sizeof(double));
    return 0;
}

int tgp_kk_twod_bootstrap(tgp_ctx *ctx, const double *x, const double *y, const double *yv, const double *yerr,
                          int64_t n, const int64_t *idx, int64_t n_boot, double min_sep, double max_sep, int nbins,
                          double *xi_out) {
    TGP_ARG(idx && xi_out && n_boot >= 1);
    {   // every index must name a point (checked before anything is launched)
        int64_t lo = 0, hi = 0;
        for (int64_t t = 0; t < n_boot * n; ++t) {
            lo = idx[t] < lo ? idx[t] : lo;
            hi = idx[t] > hi ? idx[t] : hi;
        }
        TGP_ARG(lo >= 0 && hi < n);
    }
    TGP_ARG(x && y && yv && n > 1 && nbins > 0 && nbins * nbins <= MAXB2 && max_sep > 0.0);
    {   // many resamples: one traversal of the base pairs + per-pixel pair lists (kk_boot.hip); 1 = not applicable
        const char *sw = getenv("TGP_BOOT_LISTS");
        if (!(sw && sw[0] == '0')) {
            const int rc = kk_bootstrap_lists(ctx, x, y, yv, yerr, n, idx, n_boot, min_sep, max_sep, nbins, xi_out);
            if (rc != 1) return rc;
        }
    }
    std::vector<double> acc;
    int rc = kk_run(ctx, true, x, y, yv, nullptr, yerr, n, idx, n_boot, min_sep, max_sep, nbins, acc);
    if (rc) return rc;
    const int nb = nbins * nbins;
    for (int64_t b = 0; b < n_boot; ++b) {
        const double *a = acc.data() + (size_t)b * 3 * nb;
        for (int t = 0; t < nb; ++t) xi_out[b * nb + t] = (a[nb + t] != 0.0) ? a[t] / a[nb + t] : 0.0;
    }
    return 0;
}

}  // extern "C"
