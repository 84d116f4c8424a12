// Binned 2-D statistic of scattered data (the "meanify" mean function): what
// scipy.stats.binned_statistic_2d(u, v, values, bins=[u_edges, v_edges], statistic=...) computes for
// treegp/meanify.py:76-107 -- "mean", "median", and the three "sum" passes of the weighted branch.
//
// Bin numbers follow scipy exactly (integer work, bit-exact): np.digitize against the edge arrays
// (number of edges <= x), points on the last edge folded into the last bin with scipy's rounding
// test (_bin_numbers: around(x, decimal) == around(last_edge, decimal)), everything else outside
// the edges dropped.
//
//   pass 1  one thread per point: two binary searches over edges staged in LDS, bin number kept,
//           count / sums accumulated (LDS-private histogram when it fits, else global fp64 atomics)
//   median  exclusive scan of the counts, counting-scatter of the values into per-bin segments, then
//           one workgroup per bin finds the middle order statistics by 8-bit radix selection on
//           the order-preserving integer image of the doubles: no sort, exact for any bin size
//
// HBM-bound streaming: 24-32 bytes read per point, 4 written.
#include "tgp_internal.h"

namespace {
constexpr int MAX_LDS_EDGES = 8192;       // both edge arrays together
constexpr int MAX_LDS_ACC = 12288;        // doubles of LDS histogram (96 KB)

struct BinArgs {
    const double *u, *v, *val, *err;
    int64_t n;
    const double *ue, *ve;
    int nue, nve;
    double u_last_round, v_last_round;    // np.around(last edge, decimal)
    double u_f, v_f;                      // 10^|decimal|
    int u_dec_neg, v_dec_neg;             // decimal < 0: around(x) = rint(x / f) * f
    int nacc;                             // 1: count | 2: count, sum | 3: sum w, sum w p, sum w p p
    int weighted;
};

__device__ __forceinline__ double np_around(double x, double f, int neg) {
    return neg ? __dmul_rn(rint(__ddiv_rn(x, f)), f) : __ddiv_rn(rint(__dmul_rn(x, f)), f);
}

// np.digitize(x, e) for increasing e: number of edges <= x (NaN compares false everywhere -> n, an outlier)
__device__ __forceinline__ int digitize(const double *e, int n, double x) {
    if (x != x) return n;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (e[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <bool LDS_EDGES, bool LDS_ACC>
__global__ __launch_bounds__(256) void bin_accumulate_kernel(BinArgs a, int *__restrict__ bin_out, double *__restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int nbv = a.nve - 1, nb = (a.nue - 1) * nbv;
    const double *ue = a.ue, *ve = a.ve;
    double *hist = smem;
    if (LDS_EDGES) {
        double *se = smem;
        for (int t = tid; t < a.nue; t += 256) se[t] = a.ue[t];
        for (int t = tid; t < a.nve; t += 256) se[a.nue + t] = a.ve[t];
        ue = se; ve = se + a.nue;
        hist = se + a.nue + a.nve;
    }
    if (LDS_ACC)
        for (int t = tid; t < a.nacc * nb; t += 256) hist[t] = 0.0;
    __syncthreads();
    const double ulast = a.ue[a.nue - 1], vlast = a.ve[a.nve - 1];
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < a.n; i += (int64_t)gridDim.x * 256) {
        const double x = a.u[i], y = a.v[i];
        int iu = digitize(ue, a.nue, x), iv = digitize(ve, a.nve, y);
        if (x >= ulast && np_around(x, a.u_f, a.u_dec_neg) == a.u_last_round) --iu;
        if (y >= vlast && np_around(y, a.v_f, a.v_dec_neg) == a.v_last_round) --iv;
        int b = -1;
        if (iu >= 1 && iu <= a.nue - 1 && iv >= 1 && iv <= a.nve - 1) b = (iu - 1) * nbv + (iv - 1);
        if (bin_out) bin_out[i] = b;
        if (b < 0) continue;
        double t0, t1 = 0.0, t2 = 0.0;
        if (a.weighted) {
            const double e = a.err[i], p = a.val[i];
            const double w = __ddiv_rn(1.0, __dmul_rn(e, e));          // 1.0 / params_err**2   (meanify.py:56)
            t0 = w; t1 = __dmul_rn(w, p); t2 = __dmul_rn(t1, p);        // weights * params * params   (:80)
        } else {
            t0 = 1.0;
            if (a.nacc > 1) t1 = a.val[i];
        }
        double *dst = LDS_ACC ? hist : acc;
        if (LDS_ACC) {
            __hip_atomic_fetch_add(dst + b, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (a.nacc > 1) __hip_atomic_fetch_add(dst + nb + b, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (a.nacc > 2) __hip_atomic_fetch_add(dst + 2 * nb + b, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            __hip_atomic_fetch_add(dst + b, t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.nacc > 1) __hip_atomic_fetch_add(dst + nb + b, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a.nacc > 2) __hip_atomic_fetch_add(dst + 2 * nb + b, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (LDS_ACC) {
        __syncthreads();
        for (int t = tid; t < a.nacc * nb; t += 256) {
            const double s = hist[t];
            if (s != 0.0) __hip_atomic_fetch_add(acc + t, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// offsets[b] = sum of counts[< b] (counts are doubles holding integers), one workgroup, any nb
__global__ __launch_bounds__(1024) void scan_counts_kernel(const double *__restrict__ counts, int nb, int64_t *__restrict__ offsets) {
    __shared__ int64_t part[1024];
    const int tid = threadIdx.x;
    const int per = (nb + 1023) / 1024;
    const int lo = tid * per, hi = (lo + per < nb) ? lo + per : nb;
    int64_t s = 0;
    for (int b = lo; b < hi; ++b) s += (int64_t)counts[b];
    part[tid] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        int64_t add = (tid >= o) ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += add;
        __syncthreads();
    }
    int64_t run = part[tid] - s;
    for (int b = lo; b < hi; ++b) { offsets[b] = run; run += (int64_t)counts[b]; }
    if (tid == 1023) offsets[nb] = part[1023];
}

__global__ __launch_bounds__(256) void scatter_values_kernel(const double *__restrict__ val, const int *__restrict__ bin, int64_t n,
                                                             const int64_t *__restrict__ offsets, int *__restrict__ cursor,
                                                             double *__restrict__ seg) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int b = bin[i];
        if (b < 0) continue;
        const int pos = atomicAdd(cursor + b, 1);
        seg[offsets[b] + pos] = val[i];
    }
}

// order-preserving map double -> uint64 (numbers ascending, -0.0 just below +0.0, NaNs at the ends)
__device__ __forceinline__ uint64_t ordered_key(double x) {
    const uint64_t u = (uint64_t)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(uint64_t k) {
    const uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

// key of rank r (0-based, ascending) in seg[0..cnt): MSB-first 8-bit radix selection, whole workgroup
__device__ uint64_t radix_select(const double *seg, int64_t cnt, int64_t r, unsigned *hist, uint64_t *sh_prefix, int64_t *sh_rank) {
    const int tid = threadIdx.x;
    uint64_t prefix = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        hist[tid] = 0;                                   // blockDim.x == 256
        __syncthreads();
        const uint64_t himask = (shift == 56) ? 0ull : (~0ull << (shift + 8));
        for (int64_t i = tid; i < cnt; i += 256) {
            const uint64_t k = ordered_key(seg[i]);
            if ((k & himask) == prefix) atomicAdd(hist + (unsigned)((k >> shift) & 255), 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int64_t rr = r;
            int d = 0;
            for (; d < 255; ++d) {
                if (rr < (int64_t)hist[d]) break;
                rr -= hist[d];
            }
            *sh_prefix = prefix | ((uint64_t)d << shift);
            *sh_rank = rr;
        }
        __syncthreads();
        prefix = *sh_prefix;
        r = *sh_rank;
        __syncthreads();
    }
    return prefix;
}

// scipy's median: values sorted within the bin, mid = (cnt-1)/2, (v[floor(mid)] + v[ceil(mid)]) / 2
__global__ __launch_bounds__(256) void bin_median_kernel(const double *__restrict__ seg, const int64_t *__restrict__ offsets, int nb,
                                                         double *__restrict__ out) {
    __shared__ unsigned hist[256];
    __shared__ uint64_t sh_prefix;
    __shared__ int64_t sh_rank;
    for (int b = blockIdx.x; b < nb; b += gridDim.x) {
        const int64_t o = offsets[b], cnt = offsets[b + 1] - o;
        if (cnt == 0) {
            if (threadIdx.x == 0) out[b] = __builtin_nan("");
            continue;
        }
        const int64_t lo = (cnt - 1) / 2, hi = cnt / 2;
        const uint64_t ka = radix_select(seg + o, cnt, lo, hist, &sh_prefix, &sh_rank);
        uint64_t kb = ka;
        if (hi != lo) kb = radix_select(seg + o, cnt, hi, hist, &sh_prefix, &sh_rank);
        if (threadIdx.x == 0) out[b] = __ddiv_rn(__dadd_rn(key_value(ka), key_value(kb)), 2.0);
        __syncthreads();
    }
}

// np.around's scale for scipy's "on the last edge" test: decimal = int(-log10(min edge spacing)) + 6
void around_params(const double *e, int ne, double *last_round, double *f, int *neg) {
    double dmin = e[1] - e[0];
    for (int i = 2; i < ne; ++i) dmin = (e[i] - e[i - 1] < dmin) ? e[i] - e[i - 1] : dmin;
    const int decimal = (int)(-log10(dmin)) + 6;
    *neg = decimal < 0;
    *f = pow(10.0, (double)(decimal < 0 ? -decimal : decimal));
    const double last = e[ne - 1];
    *last_round = *neg ? rint(last / *f) * *f : rint(last * *f) / *f;
}
}  // namespace

extern "C" int tgp_binned_stat_2d(tgp_ctx *ctx, const double *u, const double *v, const double *val, const double *err,
                                  int64_t n, const double *u_edges, int nu_edges, const double *v_edges, int nv_edges,
                                  int stat, double *average, double *wrms, double *count) {
    TGP_ARG(u && v && val && n > 0 && u_edges && v_edges && nu_edges >= 2 && nv_edges >= 2 && average);
    TGP_ARG(stat == TGP_STAT_MEAN || stat == TGP_STAT_MEDIAN || stat == TGP_STAT_WEIGHTED);
    TGP_ARG(stat != TGP_STAT_WEIGHTED || err);
    TGP_ARG((int64_t)(nu_edges - 1) * (nv_edges - 1) < (1ll << 30) && n < (1ll << 31));
    for (int i = 1; i < nu_edges; ++i) TGP_ARG(u_edges[i] > u_edges[i - 1]);
    for (int i = 1; i < nv_edges; ++i) TGP_ARG(v_edges[i] > v_edges[i - 1]);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int nb = (nu_edges - 1) * (nv_edges - 1);
    const bool weighted = stat == TGP_STAT_WEIGHTED, median = stat == TGP_STAT_MEDIAN;
    const int nacc = weighted ? 3 : (median ? 1 : 2);
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    const size_t need = 4 * rup(n * 8) + rup((size_t)(nu_edges + nv_edges) * 8) + rup((size_t)nacc * nb * 8) + rup(n * 4) +
                        rup((size_t)(nb + 1) * 8) + rup((size_t)nb * 4) + rup(n * 8) + rup((size_t)nb * 8);
    int rc = tgp_ensure_scratch(ctx, need);
    if (rc) return rc;
    char *base = (char *)ctx->scratch;
    size_t off = 0;
    auto take = [&](size_t b) { char *p = base + off; off += rup(b); return p; };
    double *d_u = (double *)take(n * 8), *d_v = (double *)take(n * 8), *d_val = (double *)take(n * 8),
           *d_err = (double *)take(n * 8);
    double *d_edges = (double *)take((size_t)(nu_edges + nv_edges) * 8);
    double *d_acc = (double *)take((size_t)nacc * nb * 8);
    int *d_bin = (int *)take(n * 4);
    int64_t *d_off = (int64_t *)take((size_t)(nb + 1) * 8);
    int *d_cursor = (int *)take((size_t)nb * 4);
    double *d_seg = (double *)take(n * 8);
    double *d_med = (double *)take((size_t)nb * 8);

    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    TGP_HIP(hipMemcpyAsync(d_u, u, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_v, v, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_val, val, n * 8, hipMemcpyHostToDevice, st));
    if (weighted) TGP_HIP(hipMemcpyAsync(d_err, err, n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_edges, u_edges, (size_t)nu_edges * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_edges + nu_edges, v_edges, (size_t)nv_edges * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemsetAsync(d_acc, 0, (size_t)nacc * nb * 8, st));

    BinArgs a;
    a.u = d_u; a.v = d_v; a.val = d_val; a.err = d_err; a.n = n;
    a.ue = d_edges; a.ve = d_edges + nu_edges; a.nue = nu_edges; a.nve = nv_edges;
    around_params(u_edges, nu_edges, &a.u_last_round, &a.u_f, &a.u_dec_neg);
    around_params(v_edges, nv_edges, &a.v_last_round, &a.v_f, &a.v_dec_neg);
    a.nacc = nacc; a.weighted = weighted ? 1 : 0;
    const bool lds_edges = nu_edges + nv_edges <= MAX_LDS_EDGES;
    // a private histogram costs every workgroup a flush of nacc*nb atomics: only worth it with many points per slot
    const bool lds_acc = (size_t)nacc * nb <= (size_t)MAX_LDS_ACC && n >= (int64_t)32 * nacc * nb;
    const size_t shm = ((lds_edges ? nu_edges + nv_edges : 0) + (lds_acc ? (size_t)nacc * nb : 0)) * sizeof(double);
    int64_t wg = (n + 255) / 256;
    const int64_t cap = lds_acc ? (n / ((int64_t)8 * nacc * nb) + 1) : 4096;
    if (wg > cap) wg = cap;
    if (wg > 4096) wg = 4096;
    int *bin_out = median ? d_bin : nullptr;
#define TGP_BIN_LAUNCH(E, A)                                                                                           \
    do {                                                                                                               \
        TGP_HIP(hipFuncSetAttribute((const void *)bin_accumulate_kernel<E, A>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                    (int)shm));                                                                        \
        bin_accumulate_kernel<E, A><<<(unsigned)wg, 256, shm, st>>>(a, bin_out, d_acc);                                \
    } while (0)
    if (lds_edges && lds_acc) TGP_BIN_LAUNCH(true, true);
    else if (lds_edges) TGP_BIN_LAUNCH(true, false);
    else if (lds_acc) TGP_BIN_LAUNCH(false, true);
    else TGP_BIN_LAUNCH(false, false);
#undef TGP_BIN_LAUNCH
    TGP_HIP(hipGetLastError());
    if (median) {
        scan_counts_kernel<<<1, 1024, 0, st>>>(d_acc, nb, d_off);
        TGP_HIP(hipMemsetAsync(d_cursor, 0, (size_t)nb * 4, st));
        int64_t swg = (n + 255) / 256;
        if (swg > 8192) swg = 8192;
        scatter_values_kernel<<<(unsigned)swg, 256, 0, st>>>(d_val, d_bin, n, d_off, d_cursor, d_seg);
        bin_median_kernel<<<(unsigned)(nb < 65535 ? nb : 65535), 256, 0, st>>>(d_seg, d_off, nb, d_med);
        TGP_HIP(hipGetLastError());
    }
    std::vector<double> acc((size_t)nacc * nb), med;
    TGP_HIP(hipMemcpyAsync(acc.data(), d_acc, acc.size() * 8, hipMemcpyDeviceToHost, st));
    if (median) {
        med.resize(nb);
        TGP_HIP(hipMemcpyAsync(med.data(), d_med, (size_t)nb * 8, hipMemcpyDeviceToHost, st));
    }
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[4] = ms;

    // O(bins) epilogue, in the reference's order of operations (meanify.py:95-101), no contraction
    const double nan = __builtin_nan("");
    for (int b = 0; b < nb; ++b) {
#pragma clang fp contract(off)
        if (weighted) {
            const double sw = acc[b], swp = acc[nb + b], swpp = acc[2 * nb + b];
            const double avg = swp / sw;                                   // 0/0 -> nan for an empty bin
            average[b] = avg;
            if (wrms) {
                const double t1 = (2.0 * avg) * swp;
                const double t2 = (avg * avg) * sw;
                const double wvar = (1.0 / sw) * ((swpp - t1) + t2);
                wrms[b] = sqrt(wvar);
            }
            if (count) count[b] = sw;
        } else {
            const double c = acc[b];
            average[b] = median ? med[b] : (c != 0.0 ? acc[nb + b] / c : nan);
            if (wrms) wrms[b] = 0.0;
            if (count) count[b] = c;
        }
    }
    return 0;
}
