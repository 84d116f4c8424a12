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

namespace {
constexpr int KT = 256;           // points per tile
constexpr int MAXB2 = 32 * 32;    // max TwoD pixels (nbins <= 32)
constexpr int MAXBL = 256;        // max log bins

struct KKArgs {
    const double *x, *y, *v, *w;  // v: value before mean subtraction
    const int64_t *idx;           // (n_boot, n) resample indices or nullptr
    const double *mean;           // (n_boot) mean to subtract from v (nullptr: 0)
    int64_t n;
    double min_sep, max_sep, bs, minsq, maxsq, lmin;
    int nbins, jchunks;
};

__device__ __forceinline__ void lds_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void glb_add(double *p, double v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

    const int64_t boot = blockIdx.z;
    const int64_t *idx = a.idx ? a.idx + boot * a.n : nullptr;
    const double mean = a.mean ? a.mean[boot] : 0.0;
    const int64_t ti = blockIdx.x;
    const int64_t ntile = (a.n + KT - 1) / KT;
    const int64_t i = ti * KT + tid;
    double xi = 0, yi = 0, ki = 0, wi = 0;
    const bool ivalid = i < a.n;
    if (ivalid) {
        const int64_t s = idx ? idx[i] : i;
        xi = a.x[s]; yi = a.y[s]; ki = a.v[s] - mean; wi = a.w ? a.w[s] : 1.0;
    }
    // j tiles tj >= ti, dealt round-robin over gridDim.y chunks
    for (int64_t tj = ti + blockIdx.y; tj < ntile; tj += gridDim.y) {
        __syncthreads();
        const int64_t j = tj * KT + tid;
        if (j < a.n) {
            const int64_t s = idx ? idx[j] : j;
            sx[tid] = a.x[s]; sy[tid] = a.y[s]; sk[tid] = a.v[s] - mean; sw[tid] = a.w ? a.w[s] : 1.0;
        }
        __syncthreads();
        const int cnt = (int)((a.n - tj * KT < KT) ? (a.n - tj * KT) : KT);
        const int t0 = (tj == ti) ? tid + 1 : 0;          // unordered pairs: j > i
        if (!ivalid) continue;
        for (int t = (tj == ti ? 0 : 0); t < cnt; ++t) {
            if (t < t0) continue;
            const double dx = sx[t] - xi, dy = sy[t] - yi;
            const double rsq = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
            if constexpr (TWOD) {
                const double ad = fmax(fabs(dx), fabs(dy));
                if (rsq != 0.0 && rsq >= a.minsq && ad < a.max_sep) {
                    const double ww = wi * sw[t];
                    const double wkk = ww * (ki * sk[t]);
                    const int ix = (int)(__dadd_rn(dx, a.max_sep) / a.bs), iy = (int)(__dadd_rn(dy, a.max_sep) / a.bs);
                    if (ix >= 0 && ix < a.nbins && iy >= 0 && iy < a.nbins) {
                        const int b = iy * a.nbins + ix;
                        lds_add(myh + b, wkk); lds_add(myh + nb + b, ww); lds_add(myh + 2 * nb + b, 1.0);
                    }
                    const int jx = (int)(__dadd_rn(-dx, a.max_sep) / a.bs), jy = (int)(__dadd_rn(-dy, a.max_sep) / a.bs);
                    if (jx >= 0 && jx < a.nbins && jy >= 0 && jy < a.nbins) {
                        const int b = jy * a.nbins + jx;
                        lds_add(myh + b, wkk); lds_add(myh + nb + b, ww); lds_add(myh + 2 * nb + b, 1.0);
                    }
                }
            } else {
                if (rsq >= a.minsq && rsq < a.maxsq) {
                    const double lr = 0.5 * log(rsq);
                    const int b = (int)((lr - a.lmin) / a.bs);
                    if (b >= 0 && b < a.nbins) {
                        const double ww = wi * sw[t];
                        lds_add(myh + b, ww * (ki * sk[t]));
                        lds_add(myh + nb + b, ww);
                        lds_add(myh + 2 * nb + b, ww * sqrt(rsq));
                        lds_add(myh + 3 * nb + b, ww * lr);
                        lds_add(myh + 4 * nb + b, 1.0);
                    }
                }
            }
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

static int kk_run(tgp_ctx *ctx, bool twod, const double *x, const double *y, const double *v, const double *w_host,
                  const double *yerr_host, int64_t n, const int64_t *idx, int64_t n_boot, double min_sep,
                  double max_sep, int nbins, std::vector<double> &acc_host) {
    TGP_ARG(x && y && v && n > 1 && nbins > 0 && n_boot >= 1);
    TGP_ARG(twod ? (nbins * nbins <= MAXB2) : (nbins <= MAXBL));
    TGP_ARG(max_sep > 0.0 && (twod || min_sep > 0.0));
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nacc = twod ? 3 : 5;
    const int nb = twod ? nbins * nbins : nbins;
    const size_t accb = (size_t)n_boot * nacc * nb * sizeof(double);
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = 4 * rup(n * 8) + rup(idx ? (size_t)n_boot * n * 8 : 8) + rup(n_boot * 8) + rup(accb);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return p; };
    double *d_x = (double *)take(n * 8), *d_y = (double *)take(n * 8), *d_v = (double *)take(n * 8),
           *d_w = (double *)take(n * 8);
    int64_t *d_idx = (int64_t *)take(idx ? (size_t)n_boot * n * 8 : 8);
    double *d_mean = (double *)take(n_boot * 8);
    double *d_acc = (double *)take(accb);

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
    if (idx) {
        TGP_HIP(hipMemcpyAsync(d_idx, idx, (size_t)n_boot * n * 8, hipMemcpyHostToDevice, st));
        boot_mean_kernel<<<(unsigned)n_boot, 256, 0, st>>>(d_v, d_idx, n, d_mean);
    }
    TGP_HIP(hipMemsetAsync(d_acc, 0, accb, st));

    KKArgs a;
    a.x = d_x; a.y = d_y; a.v = d_v; a.w = have_w ? d_w : nullptr;
    a.idx = idx ? d_idx : nullptr;
    a.mean = idx ? d_mean : nullptr;
    a.n = n;
    a.min_sep = min_sep; a.max_sep = max_sep;
    a.minsq = min_sep * min_sep; a.maxsq = max_sep * max_sep;
    a.nbins = nbins;
    if (twod) { a.bs = 2.0 * max_sep / nbins; a.lmin = 0.0; }
    else { a.bs = log(max_sep / min_sep) / nbins; a.lmin = log(min_sep); }
    const int64_t ntile = (n + KT - 1) / KT;
    // enough workgroups to fill the chip: split the j-tile loop when there are few i-tiles
    int jch = (int)((4096 + ntile * n_boot - 1) / (ntile * n_boot));
    if (jch < 1) jch = 1;
    if (jch > ntile) jch = (int)ntile;
    a.jchunks = jch;
    const size_t shm = (size_t)(4 * KT + 4 * nacc * nb) * sizeof(double);
    dim3 grid((unsigned)ntile, (unsigned)jch, (unsigned)n_boot), block(256);
    if (twod) {
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

int tgp_kk_twod_bootstrap(tgp_ctx *ctx, const double *x, const double *y, const double *yv, const double *yerr,
                          int64_t n, const int64_t *idx, int64_t n_boot, double min_sep, double max_sep, int nbins,
                          double *xi_out) {
    TGP_ARG(idx && xi_out && n_boot >= 1);
    for (int64_t t = 0; t < n_boot * n; ++t) TGP_ARG(idx[t] >= 0 && idx[t] < n);
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
