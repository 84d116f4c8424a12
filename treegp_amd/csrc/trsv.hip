// Triangular solves with the packed factor (scipy.linalg.cho_solve at treegp/gp_interp.py:182,
// log_likelihood.py:31), log-determinant (log_likelihood.py:33) and y.alpha (log_likelihood.py:32).
// HBM-bound: L is read once per sweep.  The 128x128 diagonal blocks were inverted by potrf128,
// so each block step is a small GEMV with W followed by a streaming GEMV over the blocks below
// (forward) or to the left (backward).
#include "tgp_internal.h"

namespace {
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// y <- W y   (W lower-triangular 128x128, ld 128), one workgroup.  Rows of W are the dot-product
// direction, so a thread-per-row read of global memory would be uncoalesced: the block is staged
// through LDS with every load in flight at once (the kernel is pure latency), then each thread
// pair takes one row from LDS.
__global__ __launch_bounds__(256) void diag_gemv_n_kernel(const double *__restrict__ W, double *y) {
    __shared__ double M[128 * 129];
    __shared__ double ys[128], part[256];
    const int tid = threadIdx.x;
    {
        const int c = (tid & 63) * 2, r0 = tid >> 6;
        double2 v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = *reinterpret_cast<const double2 *>(W + (r0 + 4 * k) * 128 + c);
        if (tid < 128) ys[tid] = y[tid];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            M[(r0 + 4 * k) * 129 + c] = v[k].x;
            M[(r0 + 4 * k) * 129 + c + 1] = v[k].y;
        }
    }
    __syncthreads();
    const int r = tid & 127, h = tid >> 7;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
    for (int c = h * 64; c < h * 64 + 64; c += 2) {
        s0 += M[r * 129 + c] * ys[c];
        s1 += M[r * 129 + c + 1] * ys[c + 1];
    }
    part[tid] = s0 + s1;
    __syncthreads();
    if (tid < 128) y[tid] = part[tid] + part[tid + 128];
}

// y <- W^T y
__global__ __launch_bounds__(256) void diag_gemv_t_kernel(const double *__restrict__ W, double *y) {
    __shared__ double ys[128];
    __shared__ double part[256];
    const int tid = threadIdx.x;
    if (tid < 128) ys[tid] = y[tid];
    __syncthreads();
    const int c = tid & 127, h = tid >> 7;
    double s = 0.0;
    for (int i = h * 64; i < h * 64 + 64; ++i) s += W[i * 128 + c] * ys[i];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) y[tid] = part[tid] + part[tid + 128];
}

// forward: y[r] -= L[r, 0:128] . z   for the rows below block kb; one wave per row
__global__ __launch_bounds__(256) void fwd_update_kernel(const double *__restrict__ Lcol, int64_t rows,
                                                         const double *__restrict__ z, double *y) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double2 zz = *reinterpret_cast<const double2 *>(z + 2 * lane);
    for (int64_t r = wv; r < rows; r += (int64_t)gridDim.x * 4) {
        const double2 a = *reinterpret_cast<const double2 *>(Lcol + r * TGP_PW + 2 * lane);
        const double s = wave_sum(a.x * zz.x + a.y * zz.y);
        if (lane == 0) y[r] -= s;
    }
}

// backward: z[128 cb + c] -= sum_r L[128 kb + r, 128 cb + c] alpha_kb[r]   for cb = blockIdx.x < kb
__global__ __launch_bounds__(256) void bwd_update_kernel(const double *__restrict__ A, int64_t Np, int kb,
                                                         const double *__restrict__ akb, double *z) {
    __shared__ double as[128];
    __shared__ double part[256];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x;
    if (tid < 128) as[tid] = akb[tid];
    __syncthreads();
    const int64_t p = cb >> 1;
    const double *blk = A + panel_off(p, Np) + ((int64_t)kb * TGP_TB - p * TGP_PW) * TGP_PW + (cb & 1) * TGP_TB;
    const int c = tid & 127, h = tid >> 7;
    double s = 0.0;
#pragma unroll 8
    for (int r = h * 64; r < h * 64 + 64; ++r) s += blk[(int64_t)r * TGP_PW + c] * as[r];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) z[(int64_t)cb * TGP_TB + tid] -= part[tid] + part[tid + 128];
}

__global__ __launch_bounds__(1024) void logdet_kernel(const double *__restrict__ A, int64_t Np, int64_t n,
                                                      double *out) {
    __shared__ double part[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t p = i >> 8;
        s += 2.0 * log(A[panel_off(p, Np) + (i - p * TGP_PW) * TGP_PW + (i & 255)]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[k];
        *out = t;
    }
}

__global__ __launch_bounds__(1024) void dot_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                   int64_t n, double *out) {
    __shared__ double part[16];
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[k];
        *out = t;
    }
}

// both scalars of a likelihood evaluation in one launch: out[0] = log det, out[1] = a . b
__global__ __launch_bounds__(1024) void logdet_dot_kernel(const double *__restrict__ A, int64_t Np, int64_t n,
                                                          const double *__restrict__ a, const double *__restrict__ b, double *out) {
    __shared__ double part[2][16];
    double s = 0.0, d = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t p = i >> 8;
        s += 2.0 * log(A[panel_off(p, Np) + (i - p * TGP_PW) * TGP_PW + (i & 255)]);
        d += a[i] * b[i];
    }
    s = wave_sum(s);
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[threadIdx.x][k];
        out[threadIdx.x] = t;
    }
}

// The right-hand side as one more row of the matrix: row Np-1 (a padding row, n < Np) <- [y (n), 0 ..., 1e300].  The
// factorisation then leaves L^-1 y in that row -- the panel solves and updates treat it like any other row below the
// diagonal -- and a likelihood evaluation needs no triangular sweep at all.  The huge diagonal keeps the row's own pivot
// positive whatever |L^-1 y|^2 is; nothing lies below or to the right of it, and the log-determinant stops at row n.
__global__ void augment_rhs_kernel(double *__restrict__ A, int64_t Np, int64_t n, const double *__restrict__ y) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Np) return;
    const int64_t p = c >> 8, r = Np - 1;
    A[panel_off(p, Np) + (r - p * TGP_PW) * TGP_PW + (c & 255)] = c < n ? y[c] : (c == r ? 1e300 : 0.0);
}
// out[0] = log det (rows < n), out[1] = |L[Np-1, 0:n]|^2
__global__ __launch_bounds__(1024) void logdet_rowsq_kernel(const double *__restrict__ A, int64_t Np, int64_t n, double *out) {
    __shared__ double part[2][16];
    double s = 0.0, d = 0.0;
    const int64_t r = Np - 1;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t p = i >> 8;
        s += 2.0 * log(A[panel_off(p, Np) + (i - p * TGP_PW) * TGP_PW + (i & 255)]);
        const double z = A[panel_off(p, Np) + (r - p * TGP_PW) * TGP_PW + (i & 255)];
        d += z * z;
    }
    s = wave_sum(s);
    d = wave_sum(d);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = s; part[1][threadIdx.x >> 6] = d; }
    __syncthreads();
    if (threadIdx.x < 2) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[threadIdx.x][k];
        out[threadIdx.x] = t;
    }
}

// z (Np) <- [L[Np-1, 0:n], 0 ...]: what the factorisation left in the augmented row, as the right-hand side of the backward sweep
__global__ void extract_row_kernel(const double *__restrict__ A, int64_t Np, int64_t n, double *__restrict__ z) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= Np) return;
    const int64_t p = c >> 8, r = Np - 1;
    z[c] = c < n ? A[panel_off(p, Np) + (r - p * TGP_PW) * TGP_PW + (c & 255)] : 0.0;
}

// b (Np) <- [y (n), 0 ...]: the padded right-hand side in one launch
__global__ void pad_copy_kernel(const double *__restrict__ y, int64_t n, int64_t Np, double *__restrict__ b) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < Np) b[i] = i < n ? y[i] : 0.0;
}

// out[c] += sign * sum_r L[r][c] a[r] over `rows` rows of a 128-column block (ld 256); partial sums
// per workgroup are combined with global fp64 atomics (out must be initialised by the caller)
__global__ __launch_bounds__(256) void gemv_t_acc_kernel(const double *__restrict__ L, int64_t rows,
                                                         const double *__restrict__ a, double *out, double sign) {
    __shared__ double part[256];
    const int tid = threadIdx.x, c = tid & 127, h = tid >> 7;
    const int64_t r0 = (int64_t)blockIdx.x * 128;
    const int64_t r1 = (r0 + 128 < rows) ? r0 + 128 : rows;
    double s = 0.0;
    for (int64_t r = r0 + h; r < r1; r += 2) s += L[r * TGP_PW + c] * a[r];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) {
        const double v = sign * (part[tid] + part[tid + 128]);
        __hip_atomic_fetch_add(out + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// multi-GPU log-determinant share: sum over the diagonal blocks this rank owns
__global__ __launch_bounds__(1024) void logdet_dist_kernel(const double *__restrict__ Aloc,
                                                           const int64_t *__restrict__ loff, int64_t nB, int64_t n,
                                                           int G, int g, double *out) {
    __shared__ double part[16];
    double s = 0.0;
    const int64_t nloc = dist_panel_blocks(0, nB, g, G);
    for (int64_t t = threadIdx.x; t < nloc * TGP_PW; t += 1024) {
        const int64_t b = dist_block_of(t >> 8, g, G);
        const int64_t r = t & 255;
        if (b * TGP_PW + r < n) s += 2.0 * log(Aloc[loff[b] + r * TGP_PW + r]);
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += part[k];
        *out = t;
    }
}

// ---- 256-row panel versions: two launches per panel per sweep -------------------------------------
// stage a 128x128 block (row stride ld) into LDS with row stride 129
__device__ __forceinline__ void stage128(const double *__restrict__ src, int ld, double *M) {
    // 32 double2 per thread, all loads in flight before the first LDS store
    const int c = (threadIdx.x & 63) * 2, r0 = threadIdx.x >> 6;
    double2 v[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) v[k] = *reinterpret_cast<const double2 *>(src + (int64_t)(r0 + 4 * k) * ld + c);
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        M[(r0 + 4 * k) * 129 + c] = v[k].x;
        M[(r0 + 4 * k) * 129 + c + 1] = v[k].y;
    }
}

// y (256) <- L_pp^-1 y with L_pp = [[L00,0],[L10,L11]]: t0 = W0 y0; y1 -= L10 t0; t1 = W1 y1.   One workgroup.
__global__ __launch_bounds__(256) void diag256_fwd_kernel(const double *__restrict__ Lpp, const double *__restrict__ W0,
                                                          const double *__restrict__ W1, double *y) {
    __shared__ double M[128 * 129];
    __shared__ double v[256], part[256];
    const int tid = threadIdx.x, r = tid & 127, h = tid >> 7;
    v[tid] = y[tid];
    stage128(W0, 128, M);
    __syncthreads();
    double s = 0.0;
    for (int c = h * 64; c < h * 64 + 64; ++c) s += M[r * 129 + c] * v[c];          // rows of W0 . y0
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[tid] = part[tid] + part[tid + 128];                               // t0
    __syncthreads();
    stage128(Lpp + (int64_t)TGP_TB * TGP_PW, TGP_PW, M);                               // L10
    __syncthreads();
    s = 0.0;
    for (int c = h * 64; c < h * 64 + 64; ++c) s += M[r * 129 + c] * v[c];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[128 + tid] -= part[tid] + part[tid + 128];
    __syncthreads();
    stage128(W1, 128, M);
    __syncthreads();
    s = 0.0;
    for (int c = h * 64; c < h * 64 + 64; ++c) s += M[r * 129 + c] * v[128 + c];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[128 + tid] = part[tid] + part[tid + 128];                         // t1
    __syncthreads();
    y[tid] = v[tid];
}

// partial[chunk][c] = sum over the chunk's rows r of L[r, c] a[r]   (rows below panel p, 256 columns)
__global__ __launch_bounds__(256) void bwd_partial_kernel(const double *__restrict__ Lrows, int64_t rows, int chunk,
                                                          const double *__restrict__ a, double *__restrict__ partial) {
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * chunk;
    const int64_t r1 = (r0 + chunk < rows) ? r0 + chunk : rows;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += Lrows[r * TGP_PW + tid] * a[r];
        s1 += Lrows[(r + 1) * TGP_PW + tid] * a[r + 1];
        s2 += Lrows[(r + 2) * TGP_PW + tid] * a[r + 2];
        s3 += Lrows[(r + 3) * TGP_PW + tid] * a[r + 3];
    }
    for (; r < r1; ++r) s0 += Lrows[r * TGP_PW + tid] * a[r];
    partial[(int64_t)blockIdx.x * TGP_PW + tid] = (s0 + s1) + (s2 + s3);
}

// y (256) <- L_pp^-T (y - sum of npart partial vectors):  t1 = W1^T y1; y0 -= L10^T t1; t0 = W0^T y0
__global__ __launch_bounds__(256) void diag256_bwd_kernel(const double *__restrict__ Lpp, const double *__restrict__ W0,
                                                          const double *__restrict__ W1, double *y,
                                                          const double *__restrict__ partial, int npart) {
    __shared__ double M[128 * 129];
    __shared__ double v[256], part[256];
    const int tid = threadIdx.x, c = tid & 127, h = tid >> 7;
    {
        double acc = 0.0;
        for (int k = 0; k < npart; ++k) acc += partial[(int64_t)k * TGP_PW + tid];     // fixed order: reproducible
        v[tid] = y[tid] - acc;
    }
    stage128(W1, 128, M);
    __syncthreads();
    double s = 0.0;
    for (int r = h * 64; r < h * 64 + 64; ++r) s += M[r * 129 + c] * v[128 + r];      // columns of W1 . y1
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[128 + tid] = part[tid] + part[tid + 128];                         // t1
    __syncthreads();
    stage128(Lpp + (int64_t)TGP_TB * TGP_PW, TGP_PW, M);                               // L10
    __syncthreads();
    s = 0.0;
    for (int r = h * 64; r < h * 64 + 64; ++r) s += M[r * 129 + c] * v[128 + r];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[tid] -= part[tid] + part[tid + 128];
    __syncthreads();
    stage128(W0, 128, M);
    __syncthreads();
    s = 0.0;
    for (int r = h * 64; r < h * 64 + 64; ++r) s += M[r * 129 + c] * v[r];
    part[tid] = s;
    __syncthreads();
    if (tid < 128) v[tid] = part[tid] + part[tid + 128];
    __syncthreads();
    y[tid] = v[tid];
}

// forward: yrows[r] -= L[r, 0:256] . z (256)   one wave per row, 2 KiB per row
__global__ __launch_bounds__(256) void fwd_update256_kernel(const double *__restrict__ Lrows, int64_t rows,
                                                            const double *__restrict__ z, double *y) {
    const int lane = threadIdx.x & 63;
    const int64_t wv = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const double2 z0 = *reinterpret_cast<const double2 *>(z + 2 * lane);
    const double2 z1 = *reinterpret_cast<const double2 *>(z + 128 + 2 * lane);
    for (int64_t r = wv; r < rows; r += (int64_t)gridDim.x * 4) {
        const double2 a = *reinterpret_cast<const double2 *>(Lrows + r * TGP_PW + 2 * lane);
        const double2 b = *reinterpret_cast<const double2 *>(Lrows + r * TGP_PW + 128 + 2 * lane);
        const double s = wave_sum(a.x * z0.x + a.y * z0.y + b.x * z1.x + b.y * z1.y);
        if (lane == 0) y[r] -= s;
    }
}

}  // namespace

// 128-block sweep (4 small launches per block): measured faster on one GPU than the fused 256-panel
// sweep below (10.6 vs 12.2 ms at N=32768); the fused kernels serve the multi-GPU driver, where fewer
// launches and collectives per block matter more.
// Big-step sweeps (trsv_big.hip) from TGP_POTRS_BIG_FROM rows on (default 2048; 0 = never); TGP_POTRS_STEP = 512 | 1024.
bool potrs_big_step(int64_t Np, int *S);
static bool potrs_big_config(int64_t Np, int *S) {
    const int64_t big_from = getenv("TGP_POTRS_BIG_FROM") ? atoll(getenv("TGP_POTRS_BIG_FROM")) : 2048;
    // step: 512 below Np = 12 288, 1024 from there.  The third doubling level of the slab build (512 -> 1024) is 2 x 2 Np 512^2
    // flops on Np / 8 tiles -- at Np = 8192 half a GPU for 0.15 ms, more than the eight extra sweep launches it saves
    // (measured 512 / 1024: Np = 2304 0.129 / 0.157 ms, 4096 0.175 / 0.224, 8192 0.384 / 0.455, 16 384 0.976 / 0.948, 32 768 2.50 / 2.46).
    const int step_env = getenv("TGP_POTRS_STEP") ? atoi(getenv("TGP_POTRS_STEP")) : (Np < 12288 ? 512 : 1024);
    *S = step_env == 512 ? 512 : 1024;
    return big_from > 0 && Np >= big_from;
}

bool potrs_big_step(int64_t Np, int *S) { return potrs_big_config(Np, S); }

// the inverse slabs of this factor: the caller's cache (built on first use) or the context's buffer (rebuilt every call).
// *need_build says whether they have to be built now.
int acquire_slabs(tgp_ctx *ctx, int64_t Np, int S, double **slab_cache, int *slab_S, double **out, bool *need_build) {
    *need_build = true;
    if (slab_cache) {
        if (*slab_cache && slab_S && *slab_S != S) {             // built for another step (TGP_POTRS_STEP changed): start over
            TGP_HIP(hipStreamSynchronize(ctx->stream));
            TGP_HIP(hipFree(*slab_cache));
            *slab_cache = nullptr;
        }
        if (*slab_cache) *need_build = false;
        else TGP_HIP(hipMalloc((void **)slab_cache, vslab_bytes(Np, S)));
        if (slab_S) *slab_S = S;
        *out = *slab_cache;
    } else {
        const size_t need = vslab_bytes(Np, S);
        if (need > ctx->vslab_bytes) {
            if (ctx->vslab) TGP_HIP(hipFree(ctx->vslab));
            ctx->vslab = nullptr;
            ctx->vslab_bytes = 0;
            TGP_HIP(hipMalloc(&ctx->vslab, need));
            ctx->vslab_bytes = need;
        }
        *out = (double *)ctx->vslab;
    }
    return 0;
}

// Tried and left off (TGP_POTRS_PIPELINE=1 turns it on): the slab build is MFMA work, the sweeps are HBM streams, and step K of
// the forward sweep needs only super-block K's slab, so the build can run on the side stream in chunks, each announced by an
// event, while the forward sweep is already under way.  Measured: N = 65 536 7.54 vs 7.53 ms, N = 32 768 3.72 vs 2.47 ms,
// N = 16 384 2.80 vs 0.93 ms, N = 8192 0.61 vs 0.46 ms -- a cross-stream event wait costs more than the chunk it hides.
static int build_slabs_pipelined(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, int S, double *slabs,
                                 SlabPipeline *pipe, bool *piped) {
    static const bool on = getenv("TGP_POTRS_PIPELINE") && atoi(getenv("TGP_POTRS_PIPELINE")) == 1;
    const int nS = (int)((Np + S - 1) / S);
    *piped = on && ctx->lookahead && !ctx->ext_stream && nS >= 8;
    if (!*piped) return launch_vslab_build(ctx, d_A, d_W, Np, S, slabs);
    int rc = tgp_ensure_side_stream(ctx);
    if (rc) return rc;
    for (auto &e : ctx->ev_slab)
        if (!e) TGP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    const int nchunk = nS >= 16 ? 8 : 4;
    pipe->chunk = (nS + nchunk - 1) / nchunk;
    hipStream_t st = ctx->stream, sd = ctx->side_stream;
    TGP_HIP(hipEventRecord(ctx->ev_slab[16], st));               // the factor is complete on the main stream
    TGP_HIP(hipStreamWaitEvent(sd, ctx->ev_slab[16], 0));
    rc = launch_vslab_build_range(ctx, st, d_A, d_W, Np, S, slabs, 0, (int64_t)pipe->chunk * S);
    if (rc) return rc;
    for (int c = 1; c * pipe->chunk < nS; ++c) {
        rc = launch_vslab_build_range(ctx, sd, d_A, d_W, Np, S, slabs, (int64_t)c * pipe->chunk * S, (int64_t)(c + 1) * pipe->chunk * S);
        if (rc) return rc;
        pipe->ready[c] = ctx->ev_slab[c];
        TGP_HIP(hipEventRecord(pipe->ready[c], sd));
    }
    return 0;
}

// the inverse slabs of a kept factor (built on first use), for code outside the sweeps (posterior covariance, likelihood
// gradient); *S = 0 when the problem is below the big-step threshold.  want_S = 0: whatever step the sweeps use (their
// cache); want_S = 1024 while the sweeps run in 512-steps (Np < 12 288): a second cache of the factor -- the block
// substitution of cov.hip is MFMA work and prefers the deeper step (10.2 against 10.9 ms at N = 8192, M = 4096).
int factor_slabs(tgp_ctx *ctx, tgp_factor *f, int want_S, int *S, const double **slabs) {
    *slabs = nullptr;
    *S = 0;
    int step;
    if (!potrs_big_config(f->Np, &step)) return 0;
    const bool own = want_S && want_S != step && !getenv("TGP_POTRS_STEP");
    if (own) step = want_S;
    double *sl = nullptr;
    bool build = false;
    int rc = acquire_slabs(ctx, f->Np, step, own ? &f->d_slabs2 : &f->d_slabs, own ? &f->slab2_S : &f->slab_S, &sl, &build);
    if (rc) return rc;
    if (build) {
        rc = launch_vslab_build(ctx, f->d_A, f->d_W, f->Np, step, sl);
        if (rc) return rc;
    }
    *S = step;
    *slabs = sl;
    return 0;
}

int launch_potrs(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b, bool forward_only,
                 double **slab_cache, int *slab_S) {
    int S;
    if (!potrs_big_config(Np, &S)) return launch_potrs_128(ctx, d_A, d_W, Np, d_b, forward_only);
    int rc = tgp_ensure_scratch2(ctx, (size_t)Np * sizeof(double));
    if (rc) return rc;
    double *slabs = nullptr;
    bool build = false, piped = false;
    rc = acquire_slabs(ctx, Np, S, slab_cache, slab_S, &slabs, &build);
    if (rc) return rc;
    SlabPipeline pipe;
    if (build) {
        rc = build_slabs_pipelined(ctx, d_A, d_W, Np, S, slabs, &pipe, &piped);
        if (rc) return rc;
    }
    return launch_potrs_big(ctx, d_A, Np, S, slabs, d_b, (double *)ctx->scratch2, forward_only, piped ? &pipe : nullptr);
}

int launch_potrs_multi(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_B, int nrhs, double **slab_cache,
                       int *slab_S) {
    int S;
    if (!potrs_big_config(Np, &S)) {
        for (int v = 0; v < nrhs; ++v) {
            int rc = launch_potrs_128(ctx, d_A, d_W, Np, d_B + (int64_t)v * Np, false);
            if (rc) return rc;
        }
        return 0;
    }
    int rc = tgp_ensure_scratch2(ctx, (size_t)nrhs * Np * sizeof(double));
    if (rc) return rc;
    double *slabs = nullptr;
    bool build = false;
    rc = acquire_slabs(ctx, Np, S, slab_cache, slab_S, &slabs, &build);
    if (rc) return rc;
    if (build) {
        rc = launch_vslab_build(ctx, d_A, d_W, Np, S, slabs);
        if (rc) return rc;
    }
    return launch_potrs_big_multi(ctx, d_A, Np, S, slabs, d_B, (double *)ctx->scratch2, nrhs);
}

int launch_potrs_128(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b, bool forward_only) {
    hipStream_t st = ctx->stream;
    const int nb = (int)(Np / TGP_TB);
    // forward: L z = b
    for (int kb = 0; kb < nb; ++kb) {
        diag_gemv_n_kernel<<<1, 256, 0, st>>>(d_W + (int64_t)kb * TGP_TB * TGP_TB, d_b + (int64_t)kb * TGP_TB);
        const int64_t rows = Np - (int64_t)(kb + 1) * TGP_TB;
        if (rows > 0) {
            const int64_t p = kb >> 1;
            const double *Lcol = d_A + panel_off(p, Np) + ((int64_t)(kb + 1) * TGP_TB - p * TGP_PW) * TGP_PW +
                                 (kb & 1) * TGP_TB;
            const unsigned g = (unsigned)((rows + 3) / 4 < 2048 ? (rows + 3) / 4 : 2048);
            fwd_update_kernel<<<g, 256, 0, st>>>(Lcol, rows, d_b + (int64_t)kb * TGP_TB,
                                                 d_b + (int64_t)(kb + 1) * TGP_TB);
        }
    }
    // backward: L^T a = z   (not needed for the quadratic form alone: y^T K^-1 y = z^T z)
    for (int kb = nb - 1; kb >= 0 && !forward_only; --kb) {
        diag_gemv_t_kernel<<<1, 256, 0, st>>>(d_W + (int64_t)kb * TGP_TB * TGP_TB, d_b + (int64_t)kb * TGP_TB);
        if (kb > 0) bwd_update_kernel<<<kb, 256, 0, st>>>(d_A, Np, kb, d_b + (int64_t)kb * TGP_TB, d_b);
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_potrs_panel256(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b) {
    hipStream_t st = ctx->stream;
    const int nP = (int)(Np / TGP_PW);
    // forward: L z = b, panel by panel (diag solve with the inverted 128-blocks, then a streaming GEMV)
    for (int p = 0; p < nP; ++p) {
        const double *Lpp = d_A + panel_off(p, Np);
        const double *W0 = d_W + (int64_t)(2 * p) * TGP_TB * TGP_TB;
        double *yp = d_b + (int64_t)p * TGP_PW;
        diag256_fwd_kernel<<<1, 256, 0, st>>>(Lpp, W0, W0 + TGP_TB * TGP_TB, yp);
        const int64_t rows = Np - (int64_t)(p + 1) * TGP_PW;
        if (rows > 0) {
            const unsigned g = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
            fwd_update256_kernel<<<g, 256, 0, st>>>(Lpp + (int64_t)TGP_PW * TGP_PW, rows, yp, yp + TGP_PW);
        }
    }
    // backward: L^T a = z; a_p = L_pp^-T (z_p - sum_{rows below} L[r, p]^T a_r), the sum as per-chunk partials
    const int chunk = 512;
    const int maxpart = (int)((Np + chunk - 1) / chunk);
    int rc = tgp_ensure_scratch2(ctx, (size_t)maxpart * TGP_PW * sizeof(double));
    if (rc) return rc;
    double *partial = (double *)ctx->scratch2;
    for (int p = nP - 1; p >= 0; --p) {
        const double *Lpp = d_A + panel_off(p, Np);
        const double *W0 = d_W + (int64_t)(2 * p) * TGP_TB * TGP_TB;
        double *yp = d_b + (int64_t)p * TGP_PW;
        const int64_t rows = Np - (int64_t)(p + 1) * TGP_PW;
        const int npart = (int)((rows + chunk - 1) / chunk);
        if (npart > 0)
            bwd_partial_kernel<<<npart, 256, 0, st>>>(Lpp + (int64_t)TGP_PW * TGP_PW, rows, chunk, yp + TGP_PW, partial);
        diag256_bwd_kernel<<<1, 256, 0, st>>>(Lpp, W0, W0 + TGP_TB * TGP_TB, yp, partial, npart);
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_logdet(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out) {
    logdet_kernel<<<1, 1024, 0, ctx->stream>>>(d_A, Np, n, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_logdet_dot(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, const double *d_a, const double *d_b, double *d_out) {
    logdet_dot_kernel<<<1, 1024, 0, ctx->stream>>>(d_A, Np, n, d_a, d_b, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_augment_rhs(tgp_ctx *ctx, double *d_A, int64_t Np, int64_t n, const double *d_y) {
    augment_rhs_kernel<<<(unsigned)((Np + 255) / 256), 256, 0, ctx->stream>>>(d_A, Np, n, d_y);
    TGP_HIP(hipGetLastError());
    return 0;
}
int launch_logdet_rowsq(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out) {
    logdet_rowsq_kernel<<<1, 1024, 0, ctx->stream>>>(d_A, Np, n, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_extract_row(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_z) {
    extract_row_kernel<<<(unsigned)((Np + 255) / 256), 256, 0, ctx->stream>>>(d_A, Np, n, d_z);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_pad_copy(tgp_ctx *ctx, const double *d_y, int64_t n, int64_t Np, double *d_b) {
    pad_copy_kernel<<<(unsigned)((Np + 255) / 256), 256, 0, ctx->stream>>>(d_y, n, Np, d_b);
    TGP_HIP(hipGetLastError());
    return 0;
}

int launch_dot(tgp_ctx *ctx, const double *d_a, const double *d_b, int64_t n, double *d_out) {
    dot_kernel<<<1, 1024, 0, ctx->stream>>>(d_a, d_b, n, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}

// ---- pieces of the block-row-cyclic triangular solves (multi-GPU driver) ---------------------
// y (256) <- L_kk^-1 y   with L_kk = [[L00, 0], [L10, L11]] given by its block (ld 256), W0 = L00^-1, W1 = L11^-1
int launch_diag256_fwd(tgp_ctx *ctx, const double *Lkk, const double *W0, const double *W1, double *y) {
    diag256_fwd_kernel<<<1, 256, 0, ctx->stream>>>(Lkk, W0, W1, y);
    TGP_HIP(hipGetLastError());
    return 0;
}
// y (256) <- L_kk^-T y
int launch_diag256_bwd(tgp_ctx *ctx, const double *Lkk, const double *W0, const double *W1, double *y, const double *s) {
    diag256_bwd_kernel<<<1, 256, 0, ctx->stream>>>(Lkk, W0, W1, y, s, s ? 1 : 0);
    TGP_HIP(hipGetLastError());
    return 0;
}
// yrows[r] -= L[r, 0:256] . z   for `rows` rows (ld 256)
int launch_fwd_update_rows(tgp_ctx *ctx, const double *Lrows, int64_t rows, const double *z, double *yrows) {
    if (rows <= 0) return 0;
    const unsigned g = (unsigned)((rows + 3) / 4 < 4096 ? (rows + 3) / 4 : 4096);
    fwd_update256_kernel<<<g, 256, 0, ctx->stream>>>(Lrows, rows, z, yrows);
    TGP_HIP(hipGetLastError());
    return 0;
}
// s (256) = sum_r L[r, 0:256]^T a[r]   (s zeroed here)
int launch_gemv_t_rows(tgp_ctx *ctx, const double *Lrows, int64_t rows, const double *a, double *s) {
    hipStream_t st = ctx->stream;
    TGP_HIP(hipMemsetAsync(s, 0, TGP_PW * sizeof(double), st));
    if (rows <= 0) return 0;
    const unsigned g = (unsigned)((rows + 127) / 128);
    gemv_t_acc_kernel<<<g, 256, 0, st>>>(Lrows, rows, a, s, 1.0);
    gemv_t_acc_kernel<<<g, 256, 0, st>>>(Lrows + TGP_TB, rows, a, s + TGP_TB, 1.0);
    TGP_HIP(hipGetLastError());
    return 0;
}
int launch_logdet_dist(tgp_ctx *ctx, const double *d_Aloc, const int64_t *d_loff, int64_t Np, int64_t n, int G, int g,
                       double *d_out) {
    logdet_dist_kernel<<<1, 1024, 0, ctx->stream>>>(d_Aloc, d_loff, Np / TGP_PW, n, G, g, d_out);
    TGP_HIP(hipGetLastError());
    return 0;
}
