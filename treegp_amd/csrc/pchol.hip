// Dataflow Cholesky for the chain-bound sizes (Np <= 16384): the whole factorisation as TWO persistent launches whose workgroups
// take tile tasks as their inputs become final, instead of a schedule of launches on two streams (chol.hip: pairs with
// look-ahead).  Same seam as chol.hip (scipy.linalg.cholesky at treegp/gp_interp.py:181, log_likelihood.py:30).
//
// Why: at N = 8192 the pair-wise schedule is work-bound in its first seven cycles -- the panel chain's kernels and the bulk
// update wait for each other's compute units (a chain kernel that takes 54 us alone takes 120 beside the bulk) -- and
// chain-bound after that; U2a runs alone on the chip between two cycles.  Here nothing waits for a launch boundary or for a
// partition of the chip: a workgroup that finishes a tile takes the most urgent tile that is ready.
//
// Right-looking by panels of 256 columns, 128 x 128 tiles; the tasks of panel k (tile rows below it: i >= 2k+2):
//   D0      potrf128 of tile (2k, 2k)                       -> L, W0          (server workgroups: 96 KB of LDS, a CU each)
//   A1 x 8  16-row slices: L10 = A10 W0^T
//   A2 x 8  16-row slices: A11 -= L10 L10^T
//   D1      potrf128 of tile (2k+1, 2k+1)                   -> L, W1          (server)
//   R12(i)  X0 = A[i][2k] W0^T, then A[i][2k+1] -= X0 L10^T
//   R3(i)   X1 = A[i][2k+1] W1^T
//   U(i,j)  A[i][j] -= X[i] X[j]^T at depth 256, 2k+2 <= j <= i            (the bulk: tickets, row-major over the triangle)
// Chain tasks carry a dependency counter and are pushed to a queue when it reaches zero (the completer of the last input pushes);
// bulk tiles of panel k are taken by ticket, and a ticket may only be claimed once the rows it needs are solved (a prefix of
// the R3 tasks), so a workgroup never holds a task whose producer might still need a workgroup -- except "the previous panel's
// update of this tile", which is held by a running workgroup by construction (tickets of panel k-1 are all claimed before
// any of panel k).  Every wait is bounded (PCHOL_SPIN_TICKS -> abort flag -> info = -7 -> the caller repeats the
// factorisation with the launch schedule).
//
// Coherence between workgroups on different XCDs WITHOUT cache-wide fences (a buffer_inv per task would throw the operand
// panels out of the XCD's L2 once a microsecond): the working matrix (d_A) is only ever read with sc0 sc1 loads and written
// with sc0 sc1 stores (coherent per instruction, as the flags are); final tiles of L go to a SECOND packed matrix (Lout) and W
// to d_W, written once with sc0 sc1 stores and read with ordinary cached loads -- a line of them cannot be in any cache
// before it is written (nobody reads it earlier, and a kernel starts with clean caches).  Completion = s_waitcnt vmcnt(0) (the
// write-through stores are acknowledged) + barrier + relaxed agent-scope atomics.  The caller swaps Lout in for d_A.
#include "tgp_internal.h"

#include "gemm_tile.h"
#ifdef TGP_POTRF_STAMPS
#undef TGP_POTRF_STAMPS          // the stamp arrays belong to chol.hip's copy of potrf128.h
#endif
#define TGP_POTRF128_BODY_ONLY
#include "potrf128.h"

namespace pc {
constexpr int AUXC = 16;                       // sc1: agent scope (17 = sc0 | sc1, system scope, is 5 - 10 x slower per access)
constexpr int MAXT = 128, MAXP = 64;
constexpr unsigned QCAP = 32768, QPCAP = 256;
constexpr unsigned long long PCHOL_SPIN_TICKS = 20000000ull;       // 0.2 s of the 100 MHz clock: a wait is microseconds

template <int AUX>
__device__ __forceinline__ double2 ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ double ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void st1(double x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, soff, AUX);
}
template <int AUX>
__device__ __forceinline__ void st2(double2 x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, x), r, voff, soff, AUX);
}

struct Queue { int avail; unsigned head, tail, pad; };
struct State {
    unsigned abort, tasks_left, bulk_panel, servers_up;
    Queue qh_q, qp_q;
    int c_D0[MAXP], c_A1[MAXP], c_A1done[MAXP], c_A2[MAXP], c_D1[MAXP];
    unsigned ticket[MAXP], total[MAXP], r3prefix[MAXP];
    int c_R12[MAXP][MAXT], c_R3[MAXP][MAXT];
    unsigned r3done[MAXP][MAXT];
    unsigned ver[MAXT][MAXT];
    unsigned qp[QPCAP];
    unsigned qh[QCAP];
};

struct TraceRec { unsigned task, wg; unsigned long long t0, t1; };
constexpr unsigned TRACE_CAP = 1u << 16;
struct Trace { unsigned n, pad[3]; TraceRec rec[TRACE_CAP]; };

struct Args {
    Trace *trace;                              // nullptr: off (TGP_PCHOL_TRACE=1: tools/dataflow_trace.py)
    double *A, *L, *W, *scratch;               // working matrix, output factor, inverse diagonal blocks, server scratch (S x 2 tiles)
    int64_t Np;
    int T, nP;
    int *info;
    int base0;                                 // (reserved)
    State *S;
};

// task word: type (3 bits) << 24 | k << 16 | i << 8 | slice
enum { T_D0 = 0, T_A1 = 1, T_A2 = 2, T_D1 = 3, T_R12 = 4, T_R3 = 5, T_U = 6 };
__device__ __forceinline__ unsigned enc(int type, int k, int i, int s) { return ((unsigned)type << 24) | ((unsigned)k << 16) | ((unsigned)i << 8) | (unsigned)s; }

__device__ __forceinline__ unsigned aload(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned aadd(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int asub(int *p) { return __hip_atomic_fetch_sub(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool acas(unsigned *p, unsigned expect, unsigned want) {
    return __hip_atomic_compare_exchange_strong(p, &expect, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Multi-producer multi-consumer queue without a compare-and-swap on the hot path (448 workgroups retrying a CAS on one word hand
// out one item per round trip: the first version of this file spent 64 of its 74 ms there): `avail` counts items that are pushed
// and not yet reserved; a consumer that decrements it from a positive value owns one item and takes the next head ticket; the
// slot of that ticket may still be a moment away (pushers finish out of order): bounded spin.
__device__ __forceinline__ void q_push(Queue *q, unsigned *slots, unsigned cap, unsigned task) {
    const unsigned t = aadd(&q->tail, 1u);
    astore(slots + (t % cap), task + 1u);      // never reused within one factorisation (cap >= tasks pushed): a zero slot is an empty one
    __hip_atomic_fetch_add(&q->avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// -1: nothing there
__device__ __forceinline__ int q_pop(Queue *q, unsigned *slots, unsigned cap, State *S);
__device__ __forceinline__ int q_pop(Queue *q, unsigned *slots, unsigned cap, State *S) {
    if (__hip_atomic_load(&q->avail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) return -1;
    if (__hip_atomic_fetch_sub(&q->avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= 0) {
        __hip_atomic_fetch_add(&q->avail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // somebody was faster: give it back
        return -1;
    }
    const unsigned h = aadd(&q->head, 1u);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned v;
    while ((v = aload(slots + (h % cap))) == 0u) {     // an earlier pusher has its ticket but has not written yet
        __builtin_amdgcn_s_sleep(1);
        if (__builtin_amdgcn_s_memrealtime() - t0 > PCHOL_SPIN_TICKS) { astore(&S->abort, 1u); return -1; }
    }
    return (int)(v - 1u);
}

// tile (i, j) of a packed lower matrix, j <= i (128 x 128, ld 256)
__device__ __forceinline__ double *tile_ptr(double *M, int64_t Np, int i, int j) {
    const int64_t pj = j >> 1;
    return M + panel_off(pj, Np) + ((int64_t)TGP_TB * i - pj * TGP_PW) * TGP_PW + (j & 1) * TGP_TB;
}

// ---- tiles ---------------------------------------------------------------------------------------------------------------------
// out (128 x 128 at cout, ld 256) = [MODE 1: cin -] A (128 x KDEPTH, ld 256) B (128 x KDEPTH, ld LDB)^T: the DTV tile of gemm_tile.h
// (A straight to VGPRs, B through LDS) with the cache policies of this file: A with A_AUX, B cached, cin coherent, cout written
// through.  Same sums in the same order as gemm_tile_dtv.
template <int MODE, int KDEPTH, int LDB, int A_AUX>
__device__ __forceinline__ void dtv_tile(const double *a_ptr, const double *b_ptr, const double *cin, double *cout) {
    constexpr int LD = TGP_PW, LSB = DTV_LSB, NW = 4, MT = 2, BPT = 4, BROWS = 32;
    double (*ldsB)[128 * LSB] = reinterpret_cast<double (*)[128 * LSB]>(dtv_lds_storage());
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t ra = tile_rsrc(a_ptr, 128 * LD * 8), rb = tile_rsrc(b_ptr, 128 * LDB * 8);
    const __amdgpu_buffer_rsrc_t rci = tile_rsrc(MODE == 1 ? cin : cout, 128 * LD * 8), rco = tile_rsrc(cout, 128 * LD * 8);
    const int va = ((16 * MT * w + l15) * LD + 2 * l4) * 8;
    const int srow = tid >> 3, kp = (tid & 7) * 2;
    const int vb = (srow * LDB + kp) * 8;
    const int vc = ((16 * MT * w + l4) * LD + l15) * 8;
    const int fb = l15 * LSB + 2 * l4;
    double2 areg[2][MT][2], rbst[BPT];
    auto load_a = [&](double2 (&dst)[MT][2], int k0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) dst[m][h] = ld2<A_AUX>(ra, va, (m * 16 * LD + k0 + 8 * h) * 8);
    };
    auto load_b = [&](int k0) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) rbst[s] = ld2<0>(rb, vb, (s * BROWS * LDB + k0) * 8);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) *reinterpret_cast<double2 *>(&ldsB[buf][(srow + BROWS * s) * LSB + kp]) = rbst[s];
    };
    load_a(areg[0], 0);
    load_b(0);
    d4 acc[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = MODE == 1 ? -ld1<AUXC>(rci, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8) : 0.0;
    store_b(0);
    __syncthreads();
    constexpr int nchunk = KDEPTH / KB;
    static_assert(nchunk % 2 == 0, "chunks in register-set pairs");
    auto step = [&](const int c, double2 (&cur)[MT][2], double2 (&nxt)[MT][2]) {
        const bool more = c + 1 < nchunk;
        if (more) {
            load_a(nxt, (c + 1) * KB);
            load_b((c + 1) * KB);
        }
        const double *Bs = ldsB[c & 1];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 bf[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) bf[n] = *reinterpret_cast<const double2 *>(&Bs[fb + n * 16 * LSB + 8 * h]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].x, bf[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].y, bf[n].y, acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_b((c + 1) & 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
        step(c, areg[0], areg[1]);
        step(c + 1, areg[1], areg[0]);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                st1<AUXC>(MODE == 1 ? -acc[m][n][r] : acc[m][n][r], rco, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8);
}

// 16 rows x 128 columns, depth 128, no LDS (nt_small_tile of gemm_tile.h with this file's cache policies):
// out (16 x 128 at cout, ld 256) = [MODE 1: cin -] a (16 x 128, ld 256, A_AUX) b (128 x 128, ld LDB, cached)^T
template <int MODE, int LDB, int A_AUX>
__device__ __forceinline__ void slice16(const double *a, const double *b, const double *cin, double *cout) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t ra = tile_rsrc(a, 16 * TGP_PW * 8);
    const int va = (l15 * TGP_PW + 2 * l4) * 8;
    const double *bp = b + (int64_t)(32 * w + l15) * LDB + 2 * l4;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
#pragma unroll 1
    for (int c0 = 0; c0 < 8; c0 += 4) {
        double2 af[4][2], bf[4][2][2];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = (c0 + u) * KB + 8 * h;
                af[u][h] = ld2<A_AUX>(ra, va, k * 8);
                bf[u][0][h] = *reinterpret_cast<const double2 *>(bp + k);
                bf[u][1][h] = *reinterpret_cast<const double2 *>(bp + (int64_t)16 * LDB + k);
            }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].x, bf[u][0][h].x, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].x, bf[u][1][h].x, acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].y, bf[u][0][h].y, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].y, bf[u][1][h].y, acc[1], 0, 0, 0);
            }
    }
    const __amdgpu_buffer_rsrc_t rci = tile_rsrc(MODE == 1 ? cin : cout, 16 * TGP_PW * 8), rco = tile_rsrc(cout, 16 * TGP_PW * 8);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int off = ((l4 + 4 * r) * TGP_PW + 32 * w + 16 * n + l15) * 8;
            if constexpr (MODE == 1) st1<AUXC>(ld1<AUXC>(rci, off, 0) - acc[n][r], rco, off, 0);
            else st1<AUXC>(acc[n][r], rco, off, 0);
        }
}

// 128 x 128 tile copy, ld_s -> ld_d, loads with LAUX, stores with SAUX (one workgroup)
template <int LAUX, int SAUX>
__device__ __forceinline__ void copy_tile(const double *src, int ld_s, double *dst, int ld_d) {
    const __amdgpu_buffer_rsrc_t rs = tile_rsrc(src, 128 * ld_s * 8), rd = tile_rsrc(dst, 128 * ld_d * 8);
    const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
    double2 v[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) v[q] = ld2<LAUX>(rs, (row * ld_s + 64 * half) * 8, q * 16);
#pragma unroll
    for (int q = 0; q < 32; ++q) st2<SAUX>(v[q], rd, (row * ld_d + 64 * half) * 8, q * 16);
}

// ---- completion: the stores of this workgroup are in memory, then thread 0 tells whoever waits ------------------------------------
__device__ __forceinline__ void task_fence() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
__device__ __forceinline__ void push_hi(State *S, unsigned t) { q_push(&S->qh_q, S->qh, QCAP, t); }
__device__ __forceinline__ void push_potrf(State *S, unsigned t) { q_push(&S->qp_q, S->qp, QPCAP, t); }

__device__ void complete(const Args &a, unsigned task) {          // thread 0 only, after task_fence()
    State *S = a.S;
    const int type = (int)(task >> 24), k = (int)((task >> 16) & 255), i = (int)((task >> 8) & 255);
    const int r0 = 2 * k + 2, T = a.T;
    switch (type) {
        case T_D0:
            if (asub(&S->c_A1[k]) == 1)
                for (int s = 0; s < 8; ++s) push_hi(S, enc(T_A1, k, 0, s));
            break;
        case T_A1:
            if (asub(&S->c_A2[k]) == 1)
                for (int s = 0; s < 8; ++s) push_hi(S, enc(T_A2, k, 0, s));
            if (asub(&S->c_A1done[k]) == 1)                    // L10 is complete: the rows below may start
                for (int r = r0; r < T; ++r)
                    if (asub(&S->c_R12[k][r]) == 1) push_hi(S, enc(T_R12, k, r, 0));
            break;
        case T_A2:
            if (asub(&S->c_D1[k]) == 1) push_potrf(S, enc(T_D1, k, 0, 0));
            break;
        case T_D1:
            for (int r = r0; r < T; ++r)
                if (asub(&S->c_R3[k][r]) == 1) push_hi(S, enc(T_R3, k, r, 0));
            break;
        case T_R12:
            if (asub(&S->c_R3[k][i]) == 1) push_hi(S, enc(T_R3, k, i, 0));
            break;
        case T_R3: {
            astore(&S->r3done[k][i], 1u);
            for (;;) {                                          // rows r0 .. r0 + prefix - 1 are solved
                const unsigned p = aload(&S->r3prefix[k]);
                if ((int)p >= T - r0 || aload(&S->r3done[k][r0 + p]) == 0u) break;
                acas(&S->r3prefix[k], p, p + 1u);
            }
            break;
        }
        case T_U: {
            const int tk = (int)(task & 255);                   // j
            astore(&S->ver[i][tk], (unsigned)(k + 1));
            if (k + 1 < a.nP && tk <= r0 + 1) {                 // a tile column of the next panel
                const int kn = k + 1;
                if (i == r0 && tk == r0) { if (asub(&S->c_D0[kn]) == 1) push_potrf(S, enc(T_D0, kn, 0, 0)); }
                else if (i == r0 + 1 && tk == r0) { if (asub(&S->c_A1[kn]) == 1) for (int s = 0; s < 8; ++s) push_hi(S, enc(T_A1, kn, 0, s)); }
                else if (i == r0 + 1 && tk == r0 + 1) { if (asub(&S->c_A2[kn]) == 1) for (int s = 0; s < 8; ++s) push_hi(S, enc(T_A2, kn, 0, s)); }
                else if (i >= r0 + 2) { if (asub(&S->c_R12[kn][i]) == 1) push_hi(S, enc(T_R12, kn, i, 0)); }
            }
            break;
        }
    }
    aadd(&S->tasks_left, 0xffffffffu);                          // -1
}

// ---- the workers --------------------------------------------------------------------------------------------------------------
// A bulk ticket of the lowest panel that still has some (fetch_add: every caller gets its own), or -1.  The ticket's rows may not be
// solved yet: the caller keeps it and serves the chain queue until they are (worker_kernel).
__device__ int take_ticket(const Args &a, unsigned &kb_out) {     // thread 0
    State *S = a.S;
    const unsigned kb = aload(&S->bulk_panel);
    if ((int)kb >= a.nP) return -1;
    const unsigned tot = aload(&S->total[kb]);
    if (aload(&S->ticket[kb]) >= tot) {
        acas(&S->bulk_panel, kb, kb + 1u);
        return -1;
    }
    // not more than the workers that could run them ahead of the solved rows: a ticket taken too early only keeps its holder
    // in the chain-serving loop
    const unsigned t = aadd(&S->ticket[kb], 1u);
    if (t >= tot) return -1;
    kb_out = kb;
    return (int)t;
}
__device__ __forceinline__ unsigned ticket_task(unsigned kb, unsigned t) {
    unsigned ai = (unsigned)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while (ai * (ai + 1u) / 2u > t) --ai;
    while ((ai + 1u) * (ai + 2u) / 2u <= t) ++ai;
    const unsigned bj = t - ai * (ai + 1u) / 2u;
    const int r0 = 2 * (int)kb + 2;
    return enc(T_U, (int)kb, r0 + (int)ai, r0 + (int)bj);
}

__global__ __launch_bounds__(256, 2) void worker_kernel(Args a) {
    __shared__ int s_task;
    State *S = a.S;
    const int tid = threadIdx.x;
    unsigned idle = 0;
    int held = -1;                                 // thread 0: a bulk ticket in hand whose rows are not solved yet (and its panel)
    unsigned held_kb = 0;
    for (;;) {
        if (tid == 0) {
            int task = -1;
            if (aload(&S->abort) != 0u) task = -9;
            else if (aload(&S->tasks_left) == 0u) task = -8;
            else {
                task = q_pop(&S->qh_q, S->qh, QCAP, S);        // the chain first
                if (task < 0) {
                    if (held < 0) held = take_ticket(a, held_kb);
                    if (held >= 0) {
                        const unsigned tt = ticket_task(held_kb, (unsigned)held);
                        const unsigned need = ((tt >> 8) & 255u) - (2u * held_kb + 2u) + 1u;      // solved rows its tile needs
                        if (aload(&S->r3prefix[held_kb]) >= need) {
                            task = (int)tt;
                            held = -1;
                        }
                    }
                }
            }
            s_task = task;
        }
        __syncthreads();
        const int task = s_task;
        __syncthreads();
        if (task == -9 || task == -8) {
            if (task == -9 && tid == 0) __hip_atomic_store(a.info, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (task < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (tid == 0 && ++idle > 2000000u) astore(&S->abort, 1u);      // ~ seconds of doing nothing: give up
            continue;
        }
        idle = 0;
        const unsigned long long tr0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const int type = task >> 24, k = (task >> 16) & 255, i = (task >> 8) & 255, s = task & 255;
        const int64_t Np = a.Np;
        const double *W0 = a.W + (int64_t)(2 * k) * TGP_TB * TGP_TB, *W1 = W0 + TGP_TB * TGP_TB;
        if (type == T_U) __builtin_amdgcn_s_setprio(0);
        else __builtin_amdgcn_s_setprio(3);
        switch (type) {
            case T_A1: {                                            // L10 slice = A10 slice W0^T -> Lout
                const double *src = tile_ptr(a.A, Np, 2 * k + 1, 2 * k) + (int64_t)16 * s * TGP_PW;
                double *dst = tile_ptr(a.L, Np, 2 * k + 1, 2 * k) + (int64_t)16 * s * TGP_PW;
                slice16<0, TGP_TB, AUXC>(src, W0, nullptr, dst);
                break;
            }
            case T_A2: {                                            // A11 slice -= L10 slice L10^T (working matrix)
                const double *l10 = tile_ptr(a.L, Np, 2 * k + 1, 2 * k);
                double *c = tile_ptr(a.A, Np, 2 * k + 1, 2 * k + 1) + (int64_t)16 * s * TGP_PW;
                slice16<1, TGP_PW, 0>(l10 + (int64_t)16 * s * TGP_PW, l10, c, c);
                break;
            }
            case T_R12: {
                double *x0 = tile_ptr(a.L, Np, i, 2 * k);
                dtv_tile<0, TGP_TB, TGP_TB, AUXC>(tile_ptr(a.A, Np, i, 2 * k), W0, nullptr, x0);     // X0 = A0 W0^T -> Lout
                task_fence();                                       // X0 is read back below (by other waves too)
                double *c = tile_ptr(a.A, Np, i, 2 * k + 1);
                dtv_tile<1, TGP_TB, TGP_PW, 0>(x0, tile_ptr(a.L, Np, 2 * k + 1, 2 * k), c, c);       // A1 -= X0 L10^T
                break;
            }
            case T_R3:
                dtv_tile<0, TGP_TB, TGP_TB, AUXC>(tile_ptr(a.A, Np, i, 2 * k + 1), W1, nullptr, tile_ptr(a.L, Np, i, 2 * k + 1));
                break;
            case T_U: {                                             // A[i][j] -= X[i] X[j]^T, depth 256 (j = s)
                if (tid == 0) {                                     // the previous panel's update of this tile has landed
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (aload(&S->ver[i][s]) != (unsigned)k) {
                        __builtin_amdgcn_s_sleep(2);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > PCHOL_SPIN_TICKS) { astore(&S->abort, 1u); break; }
                    }
                }
                __syncthreads();
                double *c = tile_ptr(a.A, Np, i, s);
                const double *xi = a.L + panel_off(k, Np) + ((int64_t)TGP_TB * i - (int64_t)k * TGP_PW) * TGP_PW;
                const double *xj = a.L + panel_off(k, Np) + ((int64_t)TGP_TB * s - (int64_t)k * TGP_PW) * TGP_PW;
                dtv_tile<1, TGP_PW, TGP_PW, 0>(xi, xj, c, c);
                break;
            }
            default: break;
        }
        task_fence();
        if (tid == 0) {
            if (a.trace) {
                const unsigned n = aadd(&a.trace->n, 1u);
                if (n < TRACE_CAP) a.trace->rec[n] = TraceRec{(unsigned)task, (unsigned)blockIdx.x, tr0, __builtin_amdgcn_s_memrealtime()};
            }
            complete(a, (unsigned)task);
        }
    }
}

// ---- the diagonal blocks: a few workgroups with potrf128's LDS image, a compute unit each ---------------------------------------
__global__ __launch_bounds__(256) void potrf_server_kernel(Args a) {
    __shared__ int s_task;
    State *S = a.S;
    const int tid = threadIdx.x;
    double *tile = a.scratch + (int64_t)blockIdx.x * 2 * TGP_TB * TGP_TB, *wtile = tile + TGP_TB * TGP_TB;
    __builtin_amdgcn_s_setprio(3);
    if (tid == 0) __hip_atomic_fetch_add(&S->servers_up, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // "I have my compute unit"
    unsigned idle = 0;
    for (;;) {
        if (tid == 0) {
            int task = -1;
            if (aload(&S->abort) != 0u) task = -9;
            else if (aload(&S->tasks_left) == 0u) task = -8;
            else task = q_pop(&S->qp_q, S->qp, QPCAP, S);
            s_task = task;
        }
        __syncthreads();
        const int task = s_task;
        __syncthreads();
        if (task == -9 || task == -8) break;
        if (task < 0) {
            __builtin_amdgcn_s_sleep(4);
            if (tid == 0 && ++idle > 8000000u) astore(&S->abort, 1u);
            continue;
        }
        idle = 0;
        const unsigned long long tr0 = a.trace ? __builtin_amdgcn_s_memrealtime() : 0ull;
        const int type = task >> 24, k = (task >> 16) & 255;
        const int d = 2 * k + (type == T_D1 ? 1 : 0);
        // the block comes into this workgroup's own scratch (coherent loads), is factored there with potrf128's ordinary accesses
        // -- nobody else ever touches the scratch -- and leaves written through: L to Lout, W to d_W
        copy_tile<AUXC, 0>(tile_ptr(a.A, a.Np, d, d), TGP_PW, tile, TGP_TB);
        task_fence();
        potrf_v2::potrf128_body<true>(potrf_v2::potrf_lds_image, tile, TGP_TB, wtile, a.info, TGP_TB * d);
        task_fence();
        copy_tile<0, AUXC>(tile, TGP_TB, tile_ptr(a.L, a.Np, d, d), TGP_PW);
        copy_tile<0, AUXC>(wtile, TGP_TB, a.W + (int64_t)d * TGP_TB * TGP_TB, TGP_TB);
        if (type == T_D0) {                                         // the block above the diagonal of the 256 x 256 block: zeros in Lout
            const __amdgpu_buffer_rsrc_t rz = tile_rsrc(tile_ptr(a.L, a.Np, d, d) + TGP_TB, 128 * TGP_PW * 8);
            const int row = tid >> 1, half = tid & 1;
#pragma unroll
            for (int q = 0; q < 32; ++q) st2<AUXC>(make_double2(0.0, 0.0), rz, (row * TGP_PW + 64 * half) * 8, q * 16);
        }
        task_fence();
        if (tid == 0) {
            if (a.trace) {
                const unsigned n = aadd(&a.trace->n, 1u);
                if (n < TRACE_CAP) a.trace->rec[n] = TraceRec{(unsigned)task, 0x10000u + (unsigned)blockIdx.x, tr0, __builtin_amdgcn_s_memrealtime()};
            }
            complete(a, (unsigned)task);
        }
    }
}

__global__ void init_kernel(State *S, int T, int nP) {
    // (the state was zeroed by a memset on the same stream)
    unsigned left = 0;
    for (int k = 0; k < nP; ++k) {
        const int m = T - 2 * k - 2;                               // tile rows below panel k
        S->c_D0[k] = k == 0 ? 0 : 1;
        S->c_A1[k] = k == 0 ? 1 : 2;
        S->c_A1done[k] = 8;
        S->c_A2[k] = k == 0 ? 8 : 9;
        S->c_D1[k] = 8;
        for (int i = 2 * k + 2; i < T; ++i) {
            S->c_R12[k][i] = k == 0 ? 1 : 3;
            S->c_R3[k][i] = 2;
        }
        S->total[k] = m > 0 ? (unsigned)(m * (m + 1) / 2) : 0u;
        left += 18u + 2u * (unsigned)(m > 0 ? m : 0) + S->total[k];
    }
    S->tasks_left = left;
    S->qp[0] = enc(T_D0, 0, 0, 0) + 1u;
    S->qp_q.tail = 1;
    S->qp_q.avail = 1;
}
}  // namespace pc

static pc::Trace *g_trace = nullptr;
static int g_launches = 0;
extern "C" int tgp_debug_pchol_launches(void) { return g_launches; }

// d_A: the packed matrix (destroyed); d_L: where the factor lands (same layout); d_W: inverse diagonal blocks.  Returns 0 when
// queued, TGP_RC_HANDOFF territory is reported through the device info word (-7) like panel_mid_kernel's time-outs.
int launch_potrf_dataflow(tgp_ctx *ctx, double *d_A, double *d_L, int64_t Np, double *d_W) {
    TGP_ARG(Np > 0 && Np % TGP_PW == 0 && Np / TGP_TB <= pc::MAXT);
    constexpr int NSERVER = 8;
    int rc = tgp_ensure_side_stream(ctx);
    if (rc) return rc;
    if (!ctx->d_pchol) {
        TGP_HIP(hipMalloc((void **)&ctx->d_pchol, sizeof(pc::State)));
        TGP_HIP(hipMalloc((void **)&ctx->d_pchol_scratch, (size_t)NSERVER * 2 * TGP_TB * TGP_TB * sizeof(double)));
    }
    hipStream_t st = ctx->stream, sd = ctx->side_stream;
    pc::State *S = (pc::State *)ctx->d_pchol;
    TGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
    TGP_HIP(hipMemsetAsync(S, 0, sizeof(pc::State), st));
    pc::Args a;
    static const bool want_trace = getenv("TGP_PCHOL_TRACE") != nullptr;
    if (want_trace && !g_trace) TGP_HIP(hipMalloc((void **)&g_trace, sizeof(pc::Trace)));
    if (g_trace) TGP_HIP(hipMemsetAsync(g_trace, 0, 16, st));
    a.trace = g_trace;
    a.A = d_A; a.L = d_L; a.W = d_W; a.scratch = (double *)ctx->d_pchol_scratch;
    a.Np = Np; a.T = (int)(Np / TGP_TB); a.nP = (int)(Np / TGP_PW);
    a.info = ctx->d_info; a.base0 = 0; a.S = S;
    pc::init_kernel<<<1, 1, 0, st>>>(S, a.T, a.nP);
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void *)pc::potrf_server_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) == hipSuccess;
    }();
    (void)attr_ok;
    // the servers first (they need a compute unit each), on the side stream; then the workers on the main stream
    TGP_HIP(hipEventRecord(ctx->ev[4], st));
    TGP_HIP(hipStreamWaitEvent(sd, ctx->ev[4], 0));
    ++g_launches;
    pc::potrf_server_kernel<<<NSERVER, 256, 128 * 1024, sd>>>(a);
    // The workers fill every slot they find: they are released only once every server sits on its compute unit (128 KB of LDS:
    // no worker fits beside it) -- launched the other way round the servers would never be placed.
    TGP_HIP(hipStreamWaitValue32(st, &S->servers_up, (unsigned)NSERVER, hipStreamWaitValueGte, 0xffffffffu));
    static const int nworkers = [] { const char *e = getenv("TGP_PCHOL_WORKERS"); return e ? atoi(e) : 2 * (256 - NSERVER); }();
    pc::worker_kernel<<<nworkers, 256, 0, st>>>(a);
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipEventRecord(ctx->ev[5], sd));
    TGP_HIP(hipStreamWaitEvent(st, ctx->ev[5], 0));
    TGP_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    ctx->timings[5] = ctx->timings[6] = ctx->timings[7] = 0.0;
    return 0;
}

// debug: the task trace of the last dataflow factorisation (TGP_PCHOL_TRACE=1): out = cap records of 4 x uint64 {task, workgroup,
// t0, t1} (100 MHz ticks); returns the number of records written, -1 when tracing is off
extern "C" int tgp_debug_pchol_trace(unsigned long long *out, int cap) {
    if (!g_trace) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    unsigned n = 0;
    if (hipMemcpy(&n, g_trace, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (n > pc::TRACE_CAP) n = pc::TRACE_CAP;
    if ((int)n > cap) n = (unsigned)cap;
    std::vector<pc::TraceRec> h(n);
    if (n && hipMemcpy(h.data(), g_trace->rec, n * sizeof(pc::TraceRec), hipMemcpyDeviceToHost) != hipSuccess) return -2;
    for (unsigned i = 0; i < n; ++i) {
        out[4 * i] = h[i].task; out[4 * i + 1] = h[i].wg; out[4 * i + 2] = h[i].t0; out[4 * i + 3] = h[i].t1;
    }
    return (int)n;
}
