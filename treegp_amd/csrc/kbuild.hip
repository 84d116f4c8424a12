// Kernel-matrix build (seam S1 of include/tgp.h).
//   kbuild_lower : K + diag(yerr^2) written straight into the packed lower panels the
//                  Cholesky consumes (treegp/gp_interp.py:180 fused with kernels.py:114-121);
//                  HBM-write-bound for the Gaussian kernel, VALU-bound for von Karman.
//   kernel_dense : dense (n, m) k(X, Y) for the host-facing kernel.__call__ replacement.
#include "tgp_internal.h"
#include "kernel_eval.h"

// One 128x128 tile per workgroup of 256 threads.  Lane l of wave w owns columns 2l, 2l+1 and
// rows w, w+4, ... of the tile, so each wave-instruction stores one full 1 KiB row segment.
// (ti, tj) are GLOBAL 128-tile coordinates; `tile` is where the tile lives (ld 256).
template <int KE>
__device__ __forceinline__ void kbuild_tile(const KParams &p, const double *__restrict__ X, int64_t n,
                                            const double *__restrict__ yerr, int64_t ti, int64_t tj,
                                            double *__restrict__ tile) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t j0 = tj * TGP_TB + 2 * lane;
    double xj0 = 0, yj0 = 0, xj1 = 0, yj1 = 0;
    if (j0 < n) { xj0 = X[2 * j0]; yj0 = X[2 * j0 + 1]; }
    if (j0 + 1 < n) { xj1 = X[2 * j0 + 2]; yj1 = X[2 * j0 + 3]; }

    const bool diag_tile = (ti == tj);
    const bool pad_tile = (ti * TGP_TB + TGP_TB > n);   // rows (and maybe columns) beyond n

#pragma unroll 4
    for (int r = wave; r < TGP_TB; r += 4) {
        const int64_t i = ti * TGP_TB + r;
        double2 v;
        if (!pad_tile || i < n) {
            const double xi = X[2 * (i < n ? i : 0)], yi = X[2 * (i < n ? i : 0) + 1];   // wave-uniform
            v.x = kernel_value<KE>(p, xi - xj0, yi - yj0);
            v.y = kernel_value<KE>(p, xi - xj1, yi - yj1);
            if (diag_tile) {
                // exact diagonal (kernels.py:121) + noise (gp_interp.py:180)
                if (i == j0) { const double e = yerr ? yerr[i] : 0.0; v.x = p.amp + e * e; }
                if (i == j0 + 1) { const double e = yerr ? yerr[i] : 0.0; v.y = p.amp + e * e; }
            }
            if (pad_tile) {
                if (j0 >= n) v.x = 0.0;
                if (j0 + 1 >= n) v.y = 0.0;
            }
        } else {
            // padded rows: identity, so the padded factor is [[L, 0], [0, I]]
            v.x = (i == j0) ? 1.0 : 0.0;
            v.y = (i == j0 + 1) ? 1.0 : 0.0;
        }
        *reinterpret_cast<double2 *>(tile + (int64_t)r * TGP_PW + 2 * lane) = v;
    }
}

template <int KE>
__global__ __launch_bounds__(256) void kbuild_lower_kernel(KParams p, const double *__restrict__ X, int64_t n,
                                                           int64_t Np, const double *__restrict__ yerr,
                                                           double *__restrict__ A) {
    // triangular tile enumeration: b -> (ti, tj), tj <= ti
    const int64_t b = blockIdx.x;
    int64_t ti = (int64_t)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
    while (ti * (ti + 1) / 2 > b) --ti;
    while ((ti + 1) * (ti + 2) / 2 <= b) ++ti;
    const int64_t tj = b - ti * (ti + 1) / 2;
    const int64_t pj = tj >> 1;
    double *tile = A + panel_off(pj, Np) + (ti * TGP_TB - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    kbuild_tile<KE>(p, X, n, yerr, ti, tj, tile);
}

// ---- single-GPU build, second generation: one workgroup = 64 consecutive rows of one 256-wide panel ------------------
// The packed layout keeps the rows of a panel back to back (2 KiB each), so a 64-row slab is ONE contiguous 128 KiB piece of
// memory and workgroup b writes bytes [128 KiB b, 128 KiB (b + 1)) of the factor: the launch is a linear stream of full
// 2 KiB row stores (two 1 KiB wave stores per row).  Lane l owns columns 2l, 2l+1, 128+2l, 128+2l+1 of the panel (coordinates
// in registers), wave w rows w, w + 4, ... (coordinates wave-uniform).  The tile above the diagonal in a panel's first 128
// rows is not part of the factor and is not written.  Gaussian kernels take their exponent pre-scaled by 0.5 log2 e and
// 2^-s from a range-check-free polynomial (17 fp64 instructions instead of libm's ~30; <= 1 ulp): with one transcendental
// per 8 bytes stored the first-generation kernel was as much VALU- as HBM-bound (2.2 ms of issue at N = 65 536).
template <int KE>
__device__ __forceinline__ double kb_value(const KParams &p, double dx, double dy, const double *tab) {
    if constexpr (KE == KE_GAUSS) return p.amp * tgp_exp2_neg(quad_form(p, dx, dy));   // p.a, p.b2, p.c pre-scaled by the host
    else return kernel_value_tab<KE>(p, dx, dy, tab);                                   // von Karman: Chebyshev table in LDS
}

__device__ __forceinline__ void kb_store(double *dst, const double2 &v) {
#ifdef TGP_KBUILD_NT
    __builtin_nontemporal_store(v.x, dst);
    __builtin_nontemporal_store(v.y, dst + 1);
#else
    *reinterpret_cast<double2 *>(dst) = v;
#endif
}

template <int KE>
__global__ __launch_bounds__(256) void kbuild_slab_kernel(KParams p, const double *__restrict__ X, int64_t n, int64_t Np,
                                                          const double *__restrict__ yerr, double *__restrict__ A) {
    constexpr int SR = 64;                                   // rows per workgroup
    __shared__ double vk_tab[KE == KE_GAUSS ? 1 : 6 * K56_NDEG];
    if constexpr (KE != KE_GAUSS) {
        vonkarman_stage_table(vk_tab);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // slabs before panel q: 4 (q P - q (q - 1) / 2) with P panels in all
    const int64_t b = blockIdx.x, P = Np / TGP_PW;
    const double B2 = 4.0 * P + 2.0;
    int64_t pj = (int64_t)((B2 - sqrt(B2 * B2 - 8.0 * (double)b)) * 0.25);
    if (pj < 0) pj = 0;
    if (pj > P - 1) pj = P - 1;
    while (pj > 0 && 4 * (pj * P - pj * (pj - 1) / 2) > b) --pj;
    while (pj + 1 < P && 4 * ((pj + 1) * P - (pj + 1) * pj / 2) <= b) ++pj;
    const int64_t slab = b - 4 * (pj * P - pj * (pj - 1) / 2);
    const int64_t i0 = pj * TGP_PW + slab * SR;              // first global row of the slab
    double *dst = A + panel_off(pj, Np) + slab * SR * TGP_PW + 2 * lane;

    int64_t jc[2] = {pj * TGP_PW + 2 * lane, pj * TGP_PW + TGP_TB + 2 * lane};
    double xj[2][2], yj[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int64_t j = jc[h] + e;
            xj[h][e] = j < n ? X[2 * j] : 0.0;
            yj[h][e] = j < n ? X[2 * j + 1] : 0.0;
        }
    const bool diag_slab = slab < TGP_PW / SR;                // rows inside the panel's 256 x 256 diagonal block
    const bool upper_half_skip = slab < TGP_TB / SR;          // first 128 rows: the second tile lies above the diagonal
    const bool pad_slab = (i0 + SR > n) || (pj * TGP_PW + TGP_PW > n);

#pragma unroll 4
    for (int r = wave; r < SR; r += 4) {
        const int64_t i = i0 + r;
        double2 v[2];
        if (!pad_slab || i < n) {
            const int64_t ic = i < n ? i : 0;
            const double xi = X[2 * ic], yi = X[2 * ic + 1];                       // wave-uniform
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h].x = kb_value<KE>(p, xi - xj[h][0], yi - yj[h][0], vk_tab);
                v[h].y = kb_value<KE>(p, xi - xj[h][1], yi - yj[h][1], vk_tab);
            }
            if (diag_slab) {
                // exact diagonal (kernels.py:121) + noise (gp_interp.py:180)
                const double e = yerr ? yerr[ic] : 0.0;
                const double d = p.amp + e * e;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (i == jc[h]) v[h].x = d;
                    if (i == jc[h] + 1) v[h].y = d;
                }
            }
            if (pad_slab) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (jc[h] >= n) v[h].x = 0.0;
                    if (jc[h] + 1 >= n) v[h].y = 0.0;
                }
            }
        } else {
            // padded rows: identity, so the padded factor is [[L, 0], [0, I]]
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v[h].x = (i == jc[h]) ? 1.0 : 0.0;
                v[h].y = (i == jc[h] + 1) ? 1.0 : 0.0;
            }
        }
        kb_store(dst + (int64_t)r * TGP_PW, v[0]);
        if (!upper_half_skip) kb_store(dst + (int64_t)r * TGP_PW + TGP_TB, v[1]);
    }
}

// multi-GPU: rank g of G owns one 256-row block per round of G (dist_block_of); blockIdx.y = local tile row,
// blockIdx.x = global tile column (tiles right of the diagonal exit).
template <int KE>
__global__ __launch_bounds__(256) void kbuild_lower_dist_kernel(KParams p, const double *__restrict__ X, int64_t n,
                                                                const double *__restrict__ yerr,
                                                                double *__restrict__ Aloc,
                                                                const int64_t *__restrict__ loff, int G, int g) {
    const int64_t lt = blockIdx.y, tj = blockIdx.x;
    const int64_t bi = dist_block_of(lt >> 1, g, G);
    const int64_t ti = 2 * bi + (lt & 1);
    if (tj > ti) return;
    const int64_t pj = tj >> 1;
    const int64_t idx = (lt >> 1) - dist_first_round(pj, g, G);       // bi among the rank's blocks >= pj
    double *tile = Aloc + loff[pj] + ((idx * TGP_PW) + (lt & 1) * TGP_TB) * TGP_PW + (tj & 1) * TGP_TB;
    kbuild_tile<KE>(p, X, n, yerr, ti, tj, tile);
}

int launch_kbuild_lower_dist(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, int64_t Np,
                             const double *d_yerr, double *d_Aloc, const int64_t *d_loff, int G, int g) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0 && G >= 1 && g >= 0 && g < G);
    const KParams p = make_kparams(k);
    const int64_t nB = Np / TGP_PW;
    const int64_t nloc = dist_panel_blocks(0, nB, g, G);
    if (nloc == 0) return 0;
    dim3 grid((unsigned)(2 * nB), (unsigned)(2 * nloc)), block(256);
    switch (ke) {
        case KE_GAUSS: kbuild_lower_dist_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_yerr, d_Aloc, d_loff, G, g); break;
        case KE_VK: kbuild_lower_dist_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_yerr, d_Aloc, d_loff, G, g); break;
        default: kbuild_lower_dist_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_yerr, d_Aloc, d_loff, G, g); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_kbuild_lower(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, int64_t Np,
                        const double *d_yerr, double *d_A) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0);
    KParams p = make_kparams(k);
    static const bool tiles = getenv("TGP_KBUILD_TILES") != nullptr;         // A/B: the first-generation 128 x 128 tile kernel
    if (tiles) {
        const int64_t T = Np / TGP_TB;
        const int64_t nt = T * (T + 1) / 2;
        dim3 grid((unsigned)nt), block(256);
        switch (ke) {
            case KE_GAUSS: kbuild_lower_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
            case KE_VK: kbuild_lower_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
            default: kbuild_lower_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
        }
        TGP_HIP(hipGetLastError());
        return 0;
    }
    const int64_t P = Np / TGP_PW;
    dim3 grid((unsigned)(4 * (P * (P + 1) / 2))), block(256);                // 64-row slabs of all panels, in memory order
    if (ke == KE_GAUSS) {
        const double c0 = 0.72134752044448170368;                            // 0.5 log2 e: exp(-q/2) = 2^-(c0 q)
        p.a *= c0;
        p.b2 *= c0;
        p.c *= c0;
    }
    switch (ke) {
        case KE_GAUSS: kbuild_slab_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
        case KE_VK: kbuild_slab_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
        default: kbuild_slab_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, Np, d_yerr, d_A); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

// dense out[i*m + j] = amp k(X_i, Y_j); self != 0: Y == X and the diagonal is exactly amp.
template <int KE>
__global__ __launch_bounds__(256) void kernel_dense_kernel(KParams p, const double *__restrict__ X, int64_t n,
                                                           const double *__restrict__ Y, int64_t m, int self,
                                                           double *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 16;
    if (j >= m) return;
    const double xj = Y[2 * j], yj = Y[2 * j + 1];
    for (int r = 0; r < 16; ++r) {
        const int64_t i = i0 + r;
        if (i >= n) break;
        double v = kernel_value<KE>(p, X[2 * i] - xj, X[2 * i + 1] - yj);
        if (self && i == j) v = p.amp;
        out[i * m + j] = v;
    }
}

int launch_kernel_dense(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_Y,
                        int64_t m, int self, double *d_out) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0);
    const KParams p = make_kparams(k);
    dim3 grid((unsigned)((m + 255) / 256), (unsigned)((n + 15) / 16)), block(256);
    switch (ke) {
        case KE_GAUSS: kernel_dense_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_Y, m, self, d_out); break;
        case KE_VK: kernel_dense_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_Y, m, self, d_out); break;
        default: kernel_dense_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_Y, m, self, d_out); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

// packed lower panels -> dense (n, n) row-major, upper part zero (tests / posterior cov)
__global__ __launch_bounds__(256) void unpack_lower_kernel(const double *__restrict__ A, int64_t Np, int64_t n,
                                                           double *__restrict__ out) {
    const int64_t j = (int64_t)blockIdx.y * 256 + threadIdx.x;
    const int64_t i = blockIdx.x;
    if (j >= n) return;
    double v = 0.0;
    if (j <= i) {
        const int64_t p = j >> 8;
        v = A[panel_off(p, Np) + (i - p * TGP_PW) * TGP_PW + (j & 255)];
    }
    out[i * n + j] = v;
}

// dense (n, n) row-major K -> packed lower panels with the noise on the diagonal and identity padding: the same 64-row
// slabs as kbuild_slab_kernel (workgroup b writes bytes [128 KiB b, 128 KiB (b + 1)) of the factor), the values read from
// the caller's matrix instead of being evaluated (reads of a row segment are contiguous too)
__global__ __launch_bounds__(256) void pack_lower_kernel(const double *__restrict__ K, int64_t n, int64_t Np,
                                                         const double *__restrict__ yerr, double *__restrict__ A) {
    constexpr int SR = 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x, P = Np / TGP_PW;
    const double B2 = 4.0 * P + 2.0;
    int64_t pj = (int64_t)((B2 - sqrt(B2 * B2 - 8.0 * (double)b)) * 0.25);
    if (pj < 0) pj = 0;
    if (pj > P - 1) pj = P - 1;
    while (pj > 0 && 4 * (pj * P - pj * (pj - 1) / 2) > b) --pj;
    while (pj + 1 < P && 4 * ((pj + 1) * P - (pj + 1) * pj / 2) <= b) ++pj;
    const int64_t slab = b - 4 * (pj * P - pj * (pj - 1) / 2);
    const int64_t i0 = pj * TGP_PW + slab * SR;
    double *dst = A + panel_off(pj, Np) + slab * SR * TGP_PW;
    for (int r = wave; r < SR; r += 4) {
        const int64_t i = i0 + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + 64 * q;
            const int64_t j = pj * TGP_PW + c;
            double v;
            if (i < n && j < n) {
                v = (j <= i) ? K[i * n + j] : K[j * n + i];          // inside the diagonal block: mirror of the lower triangle
                if (i == j) { const double e = yerr ? yerr[i] : 0.0; v += e * e; }
            } else {
                v = (i == j) ? 1.0 : 0.0;
            }
            dst[(int64_t)r * TGP_PW + c] = v;
        }
    }
}

int launch_pack_lower(tgp_ctx *ctx, const double *d_K, int64_t n, int64_t Np, const double *d_yerr, double *d_A) {
    const int64_t P = Np / TGP_PW;
    pack_lower_kernel<<<(unsigned)(4 * (P * (P + 1) / 2)), 256, 0, ctx->stream>>>(d_K, n, Np, d_yerr, d_A);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_unpack_lower(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out) {
    dim3 grid((unsigned)n, (unsigned)((n + 255) / 256)), block(256);
    unpack_lower_kernel<<<grid, block, 0, ctx->stream>>>(d_A, Np, n, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}
