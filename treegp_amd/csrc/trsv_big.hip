// Triangular sweeps in steps of S = 1024 (or 512) rows (cho_solve at treegp/gp_interp.py:182, log_likelihood.py:31).
//
// The 128-block sweep of trsv.hip is a chain of 2 N/128 dependent launches per direction (a one-workgroup GEMV with the
// inverted diagonal block, then the streaming update): 15 ms at N = 65 536 for 34 GB of factor, 28 % of HBM.  Here the
// chain has 2 N/1024 links per direction and each link streams S x (rows below) doubles:
//   1. once per factor, the inverses of the S x S diagonal blocks of L ("slabs" V, and their transposes Vt) are built
//      from the 128 x 128 inverses W that the factorisation left, by recursive doubling
//          inv [[A, 0], [B, C]] = [[A^-1, 0], [-C^-1 B A^-1, C^-1]]
//      -- three levels (128 -> 256 -> 512 -> 1024) of two MFMA products each on the tuned 128 x 128 tile of gemm_tile.h
//      plus a transposing copy (every product is "NT", so the left factor's transpose is kept alongside);
//   2. forward, super-block K:  z_K = V_K b_K (all S rows at once, one wave per row), then b[r] -= L[r, K] . z_K for the
//      rows below (one wave per row, 8 KiB per row);
//   3. backward, super-block K: a_K = Vt_K z_K, then z[c] -= sum_r L[K rows, c] a_K[r] for the columns to the left (one
//      workgroup per 128 columns: no reduction across workgroups, results do not depend on scheduling).
// Explicit inverses of diagonal blocks of L have condition numbers <= sqrt(cond K), as for the 128-blocks.
// Slab layout: Np x S as S/256 column panels of (Np, 256) row-major, so that every operand of the products is a
// 256-wide panel like the factor itself; local column lc of row r lives at panel lc >> 8, offset r * 256 + (lc & 255).
#include "tgp_internal.h"
#include "gemm_tile.h"

namespace {
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ int64_t slab_off(int64_t Np, int64_t row, int lcol) {
    return (int64_t)(lcol >> 8) * Np * 256 + row * 256 + (lcol & 255);
}

// V diagonal 128-blocks <- W, Vt diagonal 128-blocks <- W^T, and zeros in the sibling corner of each 256 x 256 diagonal block
// that the 256-deep products and the GEMVs read but nothing writes (V: upper right, Vt: lower left).  One workgroup per block.
__global__ __launch_bounds__(256) void vinit_kernel(const double *__restrict__ W, double *__restrict__ V, double *__restrict__ Vt,
                                                    int64_t Np, int S, int boff) {
    __shared__ double T[128 * 129];
    const int tid = threadIdx.x;
    const int blk = blockIdx.x + boff;               // 128-row block; a launch covers the blocks of a range of super-blocks
    const int64_t r0 = (int64_t)blk * 128;
    const int lc0 = (int)(r0 % S);
    const bool odd = (blk & 1) != 0;
    const double *src = W + r0 * 128;
    const int c = (tid & 63) * 2, rr = tid >> 6;
    const double2 zero = make_double2(0.0, 0.0);
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const int r = rr + 4 * k;
        const double2 v = *reinterpret_cast<const double2 *>(src + r * 128 + c);
        T[r * 129 + c] = v.x;
        T[r * 129 + c + 1] = v.y;
        *reinterpret_cast<double2 *>(V + slab_off(Np, r0 + r, lc0 + c)) = v;
        if (!odd) *reinterpret_cast<double2 *>(V + slab_off(Np, r0 + r, lc0 + 128 + c)) = zero;
        else *reinterpret_cast<double2 *>(Vt + slab_off(Np, r0 + r, lc0 - 128 + c)) = zero;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const int r = rr + 4 * k;
        double2 v;
        v.x = T[c * 129 + r];
        v.y = T[(c + 1) * 129 + r];
        *reinterpret_cast<double2 *>(Vt + slab_off(Np, r0 + r, lc0 + c)) = v;
    }
}

// One level of the recursive doubling, 2h-block at rows o .. o + 2h (a = first h rows, b = the next h), local column lo:
//   Tt (h x h) = Wa^T (L_ba)^T :  Tt[j, i] = sum_{k >= j} Vt[o + j, lo + k] L[o + h + i, o + k]          -> TT(o + j, lo + h + i)
//   X  (h x h) = -Wb T        :  X[i, j]  = -sum_{k <= i} V[o + h + i, lo + h + k] Tt[j, k]              -> V(o + h + i, lo + j)
// Both are NT products over k in segments of KD columns; segments that lie wholly on the zero side of the triangular factor
// (k < j resp. k > i) are skipped, which is also what keeps never-written parts of the slabs from being read.
struct Prod {
    const double *a, *b;
    double *c;
    int nseg;
    int64_t sa, sb;
    bool valid;
};
// r16: rows of C per workgroup (128 for the big tile, 16 for the latency tile)
template <int KD, bool XPROD>
__device__ __forceinline__ Prod level_prod(const double *A, double *V, const double *Vt, double *TT, int64_t Np, int S, int h, int crows,
                                           int qoff) {
    const int nti = h >> 7, ntr = h / crows;          // column tiles (128 wide), row slices
    const int t = blockIdx.x % (nti * ntr);
    const int64_t o = (int64_t)(blockIdx.x / (nti * ntr) + qoff) * 2 * h;
    const int lo = (int)(o % S);
    const int rs = (t / nti) * crows, cs = (t % nti) * 128;      // row / column offset of this tile inside the h x h block
    Prod p;
    const int nsegs = h > KD ? h / KD : 1;
    if constexpr (!XPROD) {
        // rows j = rs.., columns i = cs..
        p.valid = o + h + cs < Np;
        const int first = rs / KD;                               // k >= j
        const int64_t p0 = (o >> 8) + (int64_t)first * (KD >> 8);
        p.a = Vt + slab_off(Np, o + rs, lo + first * KD);
        p.b = A + panel_off(p0, Np) + (o + h + cs - p0 * TGP_PW) * TGP_PW + ((o + first * KD) & 255);
        p.c = TT + slab_off(Np, o + rs, lo + h + cs);
        p.nseg = nsegs - first;
        p.sa = Np * 256;
        // at most two segments (h <= 512): ONE stride suffices although consecutive panels of the factor are not equally far apart
        p.sb = panel_off(p0 + 1, Np) - panel_off(p0, Np) - (int64_t)TGP_PW * TGP_PW;
    } else {
        // rows i = rs.., columns j = cs..
        p.valid = o + h + rs < Np;
        p.a = V + slab_off(Np, o + h + rs, lo + h);
        p.b = TT + slab_off(Np, o + cs, lo + h);
        p.c = V + slab_off(Np, o + h + rs, lo + cs);
        const int last = (rs + crows - 1) / KD;                  // k <= i
        p.nseg = (last + 1 < nsegs) ? last + 1 : nsegs;
        p.sa = p.sb = Np * 256;
    }
    return p;
}

template <int KD, bool XPROD>
__global__ __launch_bounds__(256) void vprod_kernel(const double *__restrict__ A, double *V, const double *Vt, double *TT, int64_t Np,
                                                    int S, int h, int qoff) {
    const Prod p = level_prod<KD, XPROD>(A, V, Vt, TT, Np, S, h, 128, qoff);
    if (!p.valid) return;
    gemm_tile_128<XPROD ? 2 : 0, TGP_PW, KD, TileDefault, 0>(p.a, p.b, p.c, nullptr, nullptr, p.nseg, p.sa, p.sb);
}

// the same products on the latency tile (16 x 128 of C per workgroup, gemm_tile.h: nt_slice_tile): a 128 x 128 x 512 tile alone
// takes 60 - 100 us, which is what a level costs when the matrix has too few tiles to fill the chip
template <int KD, bool XPROD>
__global__ __launch_bounds__(256) void vprod_small_kernel(const double *__restrict__ A, double *V, const double *Vt, double *TT,
                                                          int64_t Np, int S, int h, int qoff) {
    const Prod p = level_prod<KD, XPROD>(A, V, Vt, TT, Np, S, h, 16, qoff);
    if (!p.valid) return;
    if (p.nseg >= 2) nt_slice_tile<XPROD ? 2 : 0, KD, 2>(nt_slice_lds_storage(), p.a, 256, p.b, 256, p.c, 256, p.a + p.sa, p.b + p.sb);
    else nt_slice_tile<XPROD ? 2 : 0, KD, 1>(nt_slice_lds_storage(), p.a, 256, p.b, 256, p.c, 256, nullptr, nullptr);
}

// Vt(o + cs + c, lo + h + rs + r) = V(o + h + rs + r, lo + cs + c) for the 128 x 128 tiles of the h x h off-diagonal block
__global__ __launch_bounds__(256) void vtrans_kernel(const double *__restrict__ V, double *__restrict__ Vt, int64_t Np, int S, int h,
                                                     int qoff) {
    __shared__ double T[128 * 129];
    const int nt = h >> 7, t = blockIdx.x % (nt * nt);
    const int64_t o = (int64_t)(blockIdx.x / (nt * nt) + qoff) * 2 * h;
    const int lo = (int)(o % S), rs = (t / nt) * 128, cs = (t % nt) * 128;
    if (o + h + rs >= Np) return;
    const int tid = threadIdx.x, c = (tid & 63) * 2, rr = tid >> 6;
    const double *src = V + slab_off(Np, o + h + rs, lo + cs);
    double *dst = Vt + slab_off(Np, o + cs, lo + h + rs);
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const int r = rr + 4 * k;
        const double2 v = *reinterpret_cast<const double2 *>(src + (int64_t)r * 256 + c);
        T[r * 129 + c] = v.x;
        T[r * 129 + c + 1] = v.y;
    }
    __syncthreads();
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
        const int r = rr + 4 * k;
        double2 v;
        v.x = T[c * 129 + r];
        v.y = T[(c + 1) * 129 + r];
        *reinterpret_cast<double2 *>(dst + (int64_t)r * 256 + c) = v;
    }
}

// All sweep kernels take NR right-hand sides at once (vector v at in / out + v * vs): the factor and the slabs are read once
// for all of them.

// out[r0 + i] = sum_lc M(r0 + i, lc) in[r0 + lc]:  M = V (lower: panels 0 .. i/256) or Vt (upper: panels i/256 .. last); one
// wave per row, whatever lies on the other side of the diagonal inside those panels is zero.  All loads of a row are
// issued together (panels outside the row's range are redirected to its diagonal panel and weighted with 0): the
// kernel is one memory latency long, and there is one of it on the critical path of every step.
template <bool UPPER, int NP, int NR>
__global__ __launch_bounds__(256) void diag_gemv_big_kernel(const double *__restrict__ M, int64_t Np, int64_t r0, int rows,
                                                            const double *__restrict__ in, double *__restrict__ out, int64_t vs) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= rows) return;
    const int pd = i >> 8, plast = (rows - 1) >> 8;
    double2 m[NP][2];
    double wgt[NP];
    int cpq[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const bool on = UPPER ? (q >= pd && q <= plast) : (q <= pd);
        cpq[q] = on ? q : pd;
        wgt[q] = on ? 1.0 : 0.0;
        const double *mp = M + (int64_t)cpq[q] * Np * 256 + (r0 + i) * 256 + 2 * lane;
        m[q][0] = *reinterpret_cast<const double2 *>(mp);
        m[q][1] = *reinterpret_cast<const double2 *>(mp + 128);
    }
#pragma unroll
    for (int v = 0; v < NR; ++v) {
        double2 x[NP][2];
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const double *xp = in + v * vs + r0 + cpq[q] * 256 + 2 * lane;
            x[q][0] = *reinterpret_cast<const double2 *>(xp);
            x[q][1] = *reinterpret_cast<const double2 *>(xp + 128);
        }
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < NP; ++q)
            s += wgt[q] * ((m[q][0].x * x[q][0].x + m[q][0].y * x[q][0].y) + (m[q][1].x * x[q][1].x + m[q][1].y * x[q][1].y));
        s = wsum(s);
        if (lane == 0) out[v * vs + r0 + i] = s;
    }
}

// forward bulk: b[r] -= L[r, columns of super-block K] . z_K   for nrows rows from row0.  One wave per row, RPW rows per
// wave with every load (the rows, z, the old b) in flight before the first use: a workgroup lives for one memory latency
template <int NP, int NR>
__global__ __launch_bounds__(256) void bulk_fwd_kernel(const double *__restrict__ A, int64_t Np, int p0, int npan, int64_t row0,
                                                       int64_t nrows, const double *__restrict__ z, double *b, int64_t vs) {
    constexpr int RPW = (16 / NP) / (NR > 1 ? 2 : 1) > 0 ? (16 / NP) / (NR > 1 ? 2 : 1) : 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t first = (int64_t)blockIdx.x * (4 * RPW) + wave * RPW;
    if (first >= nrows) return;
    double2 zr[NR][NP][2], av[RPW][NP][2];
    const int64_t mine = first + (lane % RPW);                              // lane -> (row lane % RPW, vector lane / RPW)
    const int myv = (lane / RPW) < NR ? (lane / RPW) : 0;
    const double bold = b[myv * vs + row0 + (mine < nrows ? mine : nrows - 1)];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int qq = q < npan ? q : 0;                                   // panels beyond the (short, last) super-block: weight 0 below
        const int64_t p = p0 + qq;
        const double *base = A + panel_off(p, Np) - p * TGP_PW * TGP_PW + 2 * lane;
#pragma unroll
        for (int v = 0; v < NR; ++v) {
            zr[v][q][0] = *reinterpret_cast<const double2 *>(z + v * vs + qq * 256 + 2 * lane);
            zr[v][q][1] = *reinterpret_cast<const double2 *>(z + v * vs + qq * 256 + 128 + 2 * lane);
            if (q >= npan) zr[v][q][0] = zr[v][q][1] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
            const int64_t r = first + t;
            const int64_t rr = row0 + (r < nrows ? r : nrows - 1);
            av[t][q][0] = *reinterpret_cast<const double2 *>(base + rr * TGP_PW);
            av[t][q][1] = *reinterpret_cast<const double2 *>(base + rr * TGP_PW + 128);
        }
    }
    double mysum = 0.0;
#pragma unroll
    for (int v = 0; v < NR; ++v)
#pragma unroll
        for (int t = 0; t < RPW; ++t) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < NP; ++q)
                acc += (av[t][q][0].x * zr[v][q][0].x + av[t][q][0].y * zr[v][q][0].y) +
                       (av[t][q][1].x * zr[v][q][1].x + av[t][q][1].y * zr[v][q][1].y);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);     // every lane ends up with the row's sum
            if (lane == v * RPW + t) mysum = acc;
        }
    if (lane < NR * RPW && mine < nrows) b[myv * vs + row0 + mine] = bold - mysum;
}

// backward bulk: z[c] -= sum_i L[r0 + i, c] a[r0 + i]  (i < rows) for the CW columns c0 = CW blockIdx.x .. : no reduction across
// workgroups, fixed-order combination inside (lanes, then the 8 waves through LDS).  A workgroup streams rows x CW doubles
// through ONE compute unit (~50 GB/s), so its width sets the latency of the launch: CW = 128 (1 KiB per row and wave load, 1 MiB
// per workgroup at S = 1024) when there are enough columns to fill the chip anyway, CW = 32 (four rows of 256 B per wave load,
// 256 KiB per workgroup) for the short steps, where the launch is otherwise 20 us long whatever its size.
template <int CW, int NR>
__global__ __launch_bounds__(512) void bulk_bwd_kernel(const double *__restrict__ A, int64_t Np, int64_t r0, int rows,
                                                       const double *__restrict__ a, double *z, int64_t vs) {
    constexpr int LPR = CW / 2;                     // lanes per row (a double2 each)
    constexpr int RPI = 64 / LPR;                   // rows per wave load instruction
    extern __shared__ double dyn_lds[];
    double *as = dyn_lds;                           // [NR][rows]
    double2 *part = reinterpret_cast<double2 *>(dyn_lds);                   // [NR][8][LPR], reuses the space once `as` is done with
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sub = lane / LPR, cl = lane % LPR;
    const int64_t c0 = (int64_t)blockIdx.x * CW, p = c0 >> 8;
    // wave-uniform base + 32-bit lane offsets: one address register per load in flight instead of two
    const char *ubase = reinterpret_cast<const char *>(A + panel_off(p, Np) + (r0 - p * TGP_PW) * TGP_PW + (c0 & 255));
    const int loff = 16 * cl;
    const int per = rows >> 3;                      // rows is a multiple of 256: per is a multiple of 32
    const int i0 = wave * per + sub;
    double2 zold[NR];
#pragma unroll
    for (int v = 0; v < NR; ++v) zold[v] = tid < LPR ? *reinterpret_cast<const double2 *>(z + v * vs + c0 + 2 * tid) : make_double2(0.0, 0.0);
    // two register sets of wave loads, the next set in flight while one is consumed; a set covers STEP rows, which
    // can exceed a wave's share (per = 32 with CW = 32): rows past the share are clamped for the load and weighted with 0
    constexpr int U = (CW == 128 && NR == 1) ? 16 : 8, STEP = U * RPI;
    const int end = wave * per + per;
    auto ld = [&](int i, int u) {
        const int r = i + u * RPI;
        return *reinterpret_cast<const double2 *>(ubase + (unsigned)((r < end ? r : end - 1) * (TGP_PW * 8) + loff));
    };
    double2 v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v0[u] = ld(i0, u);
#pragma unroll
    for (int v = 0; v < NR; ++v)
        for (int i = tid; i < rows; i += 512) as[v * rows + i] = a[v * vs + r0 + i];
    __syncthreads();
    auto wt = [&](int v, int i, int u) {
        const int r = i + u * RPI;
        return r < end ? as[v * rows + r] : 0.0;
    };
    double sx[NR], sy[NR];
#pragma unroll
    for (int v = 0; v < NR; ++v) sx[v] = sy[v] = 0.0;
#pragma unroll 1
    for (int i = i0; i < end; i += 2 * STEP) {
        const bool second = i - sub + STEP < end;   // wave-uniform
        if (second) {
#pragma unroll
            for (int u = 0; u < U; ++u) v1[u] = ld(i + STEP, u);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int v = 0; v < NR; ++v) {
                const double w = wt(v, i, u);
                sx[v] += v0[u].x * w;
                sy[v] += v0[u].y * w;
            }
        if (i - sub + 2 * STEP < end) {
#pragma unroll
            for (int u = 0; u < U; ++u) v0[u] = ld(i + 2 * STEP, u);
        }
        if (second) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int v = 0; v < NR; ++v) {
                    const double w = wt(v, i + STEP, u);
                    sx[v] += v1[u].x * w;
                    sy[v] += v1[u].y * w;
                }
        }
    }
    __syncthreads();                                // every wave is done with `as`
#pragma unroll
    for (int v = 0; v < NR; ++v) {
#pragma unroll
        for (int o = 32; o >= LPR; o >>= 1) {       // the RPI row groups of the wave, fixed order
            sx[v] += __shfl_xor(sx[v], o, 64);
            sy[v] += __shfl_xor(sy[v], o, 64);
        }
        if (lane < LPR) part[(v * 8 + wave) * LPR + lane] = make_double2(sx[v], sy[v]);
    }
    __syncthreads();
    if (tid < LPR) {
#pragma unroll
        for (int v = 0; v < NR; ++v) {
            double tx = 0.0, ty = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) { tx += part[(v * 8 + w) * LPR + tid].x; ty += part[(v * 8 + w) * LPR + tid].y; }
            zold[v].x -= tx;
            zold[v].y -= ty;
            *reinterpret_cast<double2 *>(z + v * vs + c0 + 2 * tid) = zold[v];
        }
    }
}
}  // namespace

size_t vslab_bytes(int64_t Np, int S) { return (size_t)2 * Np * S * sizeof(double); }

// slabs: [V | Vt], each Np x S; the build's scratch TT (Np x S) belongs to the context, not to the slabs (round 3: a kept factor
// used to carry a third of dead weight, 0.5 GB at N = 65 536).  Nothing is cleared: every part a later kernel reads is written first (vinit_kernel's
// corner blocks, the level products, the transposing copies).  Builds the slabs of rows [row_lo, row_hi) (multiples of S, or
// Np) on stream `st`: super-blocks are independent of each other.
int launch_vslab_build_range(tgp_ctx *ctx, hipStream_t st, const double *d_A, const double *d_W, int64_t Np, int S, double *slabs,
                             int64_t row_lo, int64_t row_hi) {
    const size_t tt_need = (size_t)Np * S * sizeof(double);
    if (tt_need > ctx->vslab_tt_bytes) {          // grow-only; a build in flight on another stream of this context uses the old one
        TGP_HIP(hipDeviceSynchronize());
        if (ctx->vslab_tt) TGP_HIP(hipFree(ctx->vslab_tt));
        ctx->vslab_tt = nullptr;
        ctx->vslab_tt_bytes = 0;
        TGP_HIP(hipMalloc(&ctx->vslab_tt, tt_need));
        ctx->vslab_tt_bytes = tt_need;
    }
    double *V = slabs, *Vt = slabs + Np * S, *TT = (double *)ctx->vslab_tt;
    if (row_hi > Np) row_hi = Np;
    if (row_hi <= row_lo) return 0;
    const int64_t span = row_hi - row_lo;
    vinit_kernel<<<(unsigned)(span / 128), 256, 0, st>>>(d_W, V, Vt, Np, S, (int)(row_lo / 128));
    static const int64_t small_below = getenv("TGP_VSLAB_SMALL_BELOW") ? atoll(getenv("TGP_VSLAB_SMALL_BELOW")) : 12288;
    for (int h = 128; h < S && h < Np; h *= 2) {
        const int nt = h / 128;
        const unsigned nq = (unsigned)((span + 2 * h - 1) / (2 * h));
        const int qoff = (int)(row_lo / (2 * h));
        const unsigned grid = nq * nt * nt;
        const bool small = Np < small_below && h <= 512;         // the latency tile takes at most two 256-deep segments
        const unsigned sgrid = nq * nt * (h / 16);
        if (h == 128) {
            if (small) {
                vprod_small_kernel<128, false><<<sgrid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
                vprod_small_kernel<128, true><<<sgrid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
            } else {
                vprod_kernel<128, false><<<grid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
                vprod_kernel<128, true><<<grid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
            }
        } else if (small) {
            vprod_small_kernel<256, false><<<sgrid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
            vprod_small_kernel<256, true><<<sgrid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
        } else {
            vprod_kernel<256, false><<<grid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
            vprod_kernel<256, true><<<grid, 256, 0, st>>>(d_A, V, Vt, TT, Np, S, h, qoff);
        }
        vtrans_kernel<<<grid, 256, 0, st>>>(V, Vt, Np, S, h, qoff);
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_vslab_build(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, int S, double *slabs) {
    return launch_vslab_build_range(ctx, ctx->stream, d_A, d_W, Np, S, slabs, 0, Np);
}

namespace {
template <bool UPPER, int NR>
void launch_diag_gemv(hipStream_t st, int NPmax, const double *M, int64_t Np, int64_t r0, int rows, const double *in, double *out,
                      int64_t vs) {
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (NPmax <= 2) diag_gemv_big_kernel<UPPER, 2, NR><<<grid, 256, 0, st>>>(M, Np, r0, rows, in, out, vs);
    else diag_gemv_big_kernel<UPPER, 4, NR><<<grid, 256, 0, st>>>(M, Np, r0, rows, in, out, vs);
}

template <int NP, int NR>
void launch_bulk_fwd(hipStream_t st, const double *d_A, int64_t Np, int p0, int npan, int64_t row0, int64_t below, const double *z,
                     double *b, int64_t vs) {
    constexpr int RPW = (16 / NP) / (NR > 1 ? 2 : 1) > 0 ? (16 / NP) / (NR > 1 ? 2 : 1) : 1;
    bulk_fwd_kernel<NP, NR><<<(unsigned)((below + 4 * RPW - 1) / (4 * RPW)), 256, 0, st>>>(d_A, Np, p0, npan, row0, below, z, b, vs);
}

// one step of the forward sweep: z_K = V_K b_K, then b[below] -= L[below, K] z_K
template <int NR>
void fwd_step(hipStream_t st, const double *d_A, int64_t Np, int S, const double *V, int K, double *d_b, double *d_z, int64_t vs) {
    const int NPmax = S / 256;
    const int64_t r0 = (int64_t)K * S;
    const int rows = (int)((Np - r0) < S ? (Np - r0) : S);
    launch_diag_gemv<false, NR>(st, NPmax, V, Np, r0, rows, d_b, d_z, vs);
    const int64_t below = Np - (r0 + rows);
    if (below > 0) {
        const int p0 = (int)(r0 / 256), npan = rows / 256;
        if (NPmax <= 2) launch_bulk_fwd<2, NR>(st, d_A, Np, p0, npan, r0 + rows, below, d_z + r0, d_b, vs);
        else launch_bulk_fwd<4, NR>(st, d_A, Np, p0, npan, r0 + rows, below, d_z + r0, d_b, vs);
    }
}

// NR right-hand sides (vectors at stride vs in d_b and d_z) through both sweeps.  `backward_only`: d_z already holds L^-1 b
// (it came out of the factorisation as the augmented row, api.hip).
template <int NR>
int potrs_big_nr(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_b, double *d_z, int64_t vs,
                 bool forward_only, const SlabPipeline *pipe = nullptr, bool backward_only = false) {
    hipStream_t st = ctx->stream;
    const double *V = slabs, *Vt = slabs + Np * S;
    const int nS = (int)((Np + S - 1) / S);
    const int NPmax = S / 256;
    for (int K = 0; K < nS && !backward_only; ++K) {
        // slabs built beside this sweep (launch_potrs): super-block K's chunk has to be there
        if (pipe && K > 0 && K % pipe->chunk == 0) TGP_HIP(hipStreamWaitEvent(st, pipe->ready[K / pipe->chunk], 0));
        fwd_step<NR>(st, d_A, Np, S, V, K, d_b, d_z, vs);
    }
    if (forward_only) {
        for (int v = 0; v < NR; ++v)
            TGP_HIP(hipMemcpyAsync(d_b + v * vs, d_z + v * vs, (size_t)Np * sizeof(double), hipMemcpyDeviceToDevice, st));
        TGP_HIP(hipGetLastError());
        return 0;
    }
    for (int K = nS - 1; K >= 0; --K) {
        const int64_t r0 = (int64_t)K * S;
        const int rows = (int)((Np - r0) < S ? (Np - r0) : S);
        launch_diag_gemv<true, NR>(st, NPmax, Vt, Np, r0, rows, d_z, d_b, vs);
        if (K > 0) {
            if (r0 / 128 >= 512) {
                const size_t lds = (size_t)NR * (rows * 8 > 8 * 64 * 16 ? rows * 8 : 8 * 64 * 16);
                bulk_bwd_kernel<128, NR><<<(unsigned)(r0 / 128), 512, lds, st>>>(d_A, Np, r0, rows, d_b, d_z, vs);
            } else {
                const size_t lds = (size_t)NR * (rows * 8 > 8 * 16 * 16 ? rows * 8 : 8 * 16 * 16);
                bulk_bwd_kernel<32, NR><<<(unsigned)(r0 / 32), 512, lds, st>>>(d_A, Np, r0, rows, d_b, d_z, vs);
            }
        }
    }
    TGP_HIP(hipGetLastError());
    return 0;
}
}  // namespace

// d_b (Np) <- L^-T L^-1 d_b (or L^-1 d_b when forward_only) with the slabs of this factor; d_z: Np doubles of scratch
int launch_potrs_big(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_b, double *d_z,
                     bool forward_only, const SlabPipeline *pipe) {
    return potrs_big_nr<1>(ctx, d_A, Np, S, slabs, d_b, d_z, Np, forward_only, pipe);
}

// the backward sweep alone: d_z already holds L^-1 b (the augmented row of the factorisation, api.hip)
int launch_potrs_big_bwd(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_b, double *d_z) {
    return potrs_big_nr<1>(ctx, d_A, Np, S, slabs, d_b, d_z, Np, false, nullptr, true);
}

// nrhs right-hand sides, rows of d_B (nrhs, Np), in groups of 4, 2, 1; d_Z: scratch of the same shape
int launch_potrs_big_multi(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_B, double *d_Z, int nrhs) {
    int v = 0, rc = 0;
    for (; v + 4 <= nrhs && !rc; v += 4) rc = potrs_big_nr<4>(ctx, d_A, Np, S, slabs, d_B + (int64_t)v * Np, d_Z + (int64_t)v * Np, Np, false);
    for (; v + 2 <= nrhs && !rc; v += 2) rc = potrs_big_nr<2>(ctx, d_A, Np, S, slabs, d_B + (int64_t)v * Np, d_Z + (int64_t)v * Np, Np, false);
    for (; v < nrhs && !rc; ++v) rc = potrs_big_nr<1>(ctx, d_A, Np, S, slabs, d_B + (int64_t)v * Np, d_Z + (int64_t)v * Np, Np, false);
    return rc;
}
