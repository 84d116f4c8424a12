// Blocked right-looking fp64 Cholesky on the packed lower panels (seam S2 of include/tgp.h;
// replaces scipy.linalg.cholesky at treegp/gp_interp.py:181 and log_likelihood.py:30).
//
// F(k), factor one 256-wide panel (two 128-column halves):
//   potrf128  diag block 0     one workgroup: L11 in place + W11 = L11^-1          (potrf128.h)
//   gemm<0>   rows below       X = A W11^T            (triangular solve as a GEMM, in place)
//   gemm<1>   column half 1    A[:,128:256] -= X X_d^T (depth 128)
//   potrf128  diag block 1     L22, W22
//   gemm<0>   rows below       X = A W22^T
// Schedule (launch_potrf): panels in groups of four (pairs below N = 18432): inside a group every panel is first
// brought up to date by one strip launch against the panels before it, then factored; everything to the right of
// the group is updated in ONE pass of depth 1024 (512) -- that is where the N^3/3 flops are -- split so that the
// next group is factored on a priority side stream underneath it.
// The GEMMs are the NT tiles of gemm_tile.h on v_mfma_f64_16x16x4_f64 (trailing update: the DTV tile, A straight
// into VGPRs; panel solves: the 2x2-wave tile; few-tile steps: the latency tile); two workgroups per CU.
#include "tgp_internal.h"

#include "gemm_tile.h"
#include "potrf128.h"

#ifdef TGP_POTRF_STAMPS
__device__ unsigned long long tgp_gemm_stamps[1024 * 4];
__device__ int tgp_gemm_stamp_grid = 43;
__device__ unsigned long long tgp_queue_stamps[1024 * 4];      // queued bulk update with T == tgp_queue_stamp_T: per workgroup
__device__ int tgp_queue_stamp_T = 40;
__device__ unsigned long long tgp_mid_stamps[25 * 8];          // panel_mid_kernel of the panel whose first row is tgp_mid_stamp_base: workgroups 0 .. 24
__device__ int tgp_mid_stamp_base = 5120;
#define TGP_MID_STAMP(i)                                                                                   \
    do {                                                                                                   \
        if (threadIdx.x == 0 && base == tgp_mid_stamp_base && blockIdx.x < 25) tgp_mid_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define TGP_MID_STAMP(i) ((void)0)
#endif

namespace {
// a column of 128-row tiles: tile t uses A rows [128 t, +128), the fixed B block, C rows [128 t, +128)
template <int MODE, int LDB>
__global__ __launch_bounds__(256, 2) void gemm_col_kernel(const double *A, const double *B, double *C) {
    const int64_t t = blockIdx.x;
    TGP_CHAIN_PRIO();
#ifdef TGP_POTRF_STAMPS
    const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
#endif
    gemm_tile_128<MODE, LDB, TGP_TB>(A + t * 128 * TGP_PW, B, C + t * 128 * TGP_PW);
#ifdef TGP_POTRF_STAMPS
    if (MODE == 0 && threadIdx.x == 0 && (int)gridDim.x == tgp_gemm_stamp_grid) {      // one chosen call: when and where every workgroup ran
        tgp_gemm_stamps[blockIdx.x * 4 + 0] = t_in;
        tgp_gemm_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
        tgp_gemm_stamps[blockIdx.x * 4 + 2] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 8) | __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
    }
#endif
}

// The same two jobs on the latency tile (nt_small_tile): 16-row slices, for steps with only a few tiles.
template <int MODE, int LDB>
__global__ __launch_bounds__(256) void gemm_col_small_kernel(const double *A, const double *B, double *C) {
    const int64_t o = (int64_t)blockIdx.x * 16 * TGP_PW;
    TGP_CHAIN_PRIO();
    nt_slice_tile<MODE, TGP_TB, 1>(nt_slice_lds_storage(), A + o, TGP_PW, B, LDB, C + o, TGP_PW, nullptr, nullptr);
}
// Rows below a factored 256 x 256 diagonal block, all three steps of their solve in one launch: X0 = R0 W0^T,
// R1 -= X0 L10^T, X1 = R1 W1^T (R0 | R1 = the two 128-column halves of the rows, L10 = rows 128..255 of the block's first
// half, W0 / W1 the inverted 128-blocks).  A row tile depends on nothing but itself and the diagonal block.  Used by the
// multi-GPU panel chain (tgp_dd_trsm), where every launch beside the bulk update waits for workgroup slots: one wait
// instead of three.  (On one GPU the same fusion did not pay: same number of launches on the chain, see LAB_NOTES.md A.4.)
// (keep != nullptr: the grid also copies the two inverted blocks, 2 x 128 x 128 doubles at W0, to `keep` -- a rank that received
// them in the broadcast keeps them for the solves; as a copy of its own it was one more 5 us kernel on the panel chain)
__device__ __forceinline__ void keep_w_share(double *__restrict__ keep, const double *__restrict__ W0) {
    if (!keep) return;
    constexpr int n2 = TGP_TB * TGP_TB;                        // double2 elements
    const int per = (n2 + (int)gridDim.x - 1) / (int)gridDim.x;
    const int lo = (int)blockIdx.x * per, hi = lo + per < n2 ? lo + per : n2;
    for (int i = lo + (int)threadIdx.x; i < hi; i += 256)
        reinterpret_cast<double2 *>(keep)[i] = reinterpret_cast<const double2 *>(W0)[i];
}
__global__ __launch_bounds__(256, 2) void panel_tall_kernel(double *rows0, const double *W0, const double *L10, const double *W1,
                                                            double *keep) {
    double *rows = rows0 + (int64_t)blockIdx.x * 128 * TGP_PW;
    TGP_CHAIN_PRIO();
    keep_w_share(keep, W0);
    gemm_tile_128<0, TGP_TB, TGP_TB>(rows, W0, rows);
    __syncthreads();
    gemm_tile_128<1, TGP_PW, TGP_TB>(rows, L10, rows + TGP_TB);
    __syncthreads();
    gemm_tile_128<0, TGP_TB, TGP_TB>(rows + TGP_TB, W1, rows + TGP_TB);
}

// The same three steps on the latency tile, one workgroup per 16-row slice (8 x the workgroups, each ~1/5 of the time): for a
// rank's panel solves of a multi-GPU factorisation, which hold 2 - 64 row tiles and sit on the panel chain's critical path
// (59 us per panel in the 128-row form whatever the number of tiles, kernel trace of a rank's share at 8 ranks).
__global__ __launch_bounds__(256) void panel_tall_small_kernel(double *rows0, const double *W0, const double *L10, const double *W1,
                                                               double *keep) {
    double *rows = rows0 + (int64_t)blockIdx.x * 16 * TGP_PW;
    TGP_CHAIN_PRIO();
    keep_w_share(keep, W0);
    double *lds = nt_slice_lds_storage();
    nt_slice_tile<0, TGP_TB, 1>(lds, rows, TGP_PW, W0, TGP_TB, rows, TGP_PW, nullptr, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's X0 is in the L2 before the others read it
    __syncthreads();
    nt_slice_tile<1, TGP_TB, 1>(lds, rows, TGP_PW, L10, TGP_PW, rows + TGP_TB, TGP_PW, nullptr, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    nt_slice_tile<0, TGP_TB, 1>(lds, rows + TGP_TB, TGP_PW, W1, TGP_TB, rows + TGP_TB, TGP_PW, nullptr, nullptr);
}

// rows 128..255 of a 256 x 256 diagonal block between its two potrf128 calls, one workgroup, one launch:
// L10 = A10 W0^T, then A11 -= L10 L10^T
__global__ __launch_bounds__(256, 2) void diag_mid_kernel(double *R1, const double *W0) {
    TGP_CHAIN_PRIO();
    gemm_tile_128<0, TGP_TB, TGP_TB>(R1, W0, R1);
    __syncthreads();
    gemm_tile_128<1, TGP_PW, TGP_TB>(R1, R1, R1 + TGP_TB);
}

// ---- the step between the two diagonal blocks of a panel as ONE launch with in-kernel hand-offs -------------------------------
// Between potrf128(0,0) and potrf128(1,1) the chain used to run two launches over ALL rows below (X0 = A0 W0^T, then
// A1 -= X0 L10^T), 24 + 25 us at N = 8192, although the second diagonal block needs only its own 128 rows of them.  Here the
// workgroups of one launch are, in dispatch order (every wait is on workgroups with LOWER indices, which the dispatcher has
// started before: progress never depends on how many workgroups are resident):
//   0 ..  7   A1: sixteen-row slices of rows 128..255:  L10 = A10 W0^T (in place: whole slices)            -> count sync[0]
//   8 .. 23   A2: the same slices as two column halves each, after sync[0] == 8:  A11 -= L10 L10^T         -> count sync[1]
//   24        after sync[1] == 16: potrf128 of block (1,1): L11 in place, W1 = L11^-1
//   25 ..     the rows below the 256 x 256 block (128-row tiles, or 16-row slices where SMALLROWS): X0 = A0 W0^T, then -- after
//             sync[0] == 8 -- A1 -= X0 L10^T, both under the diagonal block's 25 us
// (Second session of round 5, in-kernel stamps tools/mid_stamps.py: from the end of the first diagonal block to the start of the
// second one's body 18.2 us with eight whole slices per step on the direct-from-global latency tile, 14.2 us on the LDS-staged
// one, ~12 us with the second step's slices in column halves: profiles/r05_mid_stamps.txt.)
// The third product of the rows (X1 = A1 W1^T) stays a launch of its own behind this one: folded in, its workgroups would
// hold their compute units spinning for W1.  Hand-off: the producer's waves wait for their stores (vmcnt(0)), barrier, one
// release fence at agent scope (writes the XCD's L2 back), one relaxed atomic increment; the consumer's thread 0 polls with
// relaxed agent-scope loads, then one acquire fence (invalidates the XCD's L2 and the CU's L1), barrier.
// EVERY SPIN IS BOUNDED: after PANEL_SPIN_TICKS of the 100 MHz clock (50 ms; a wait is a few microseconds) the waiter raises
// sync[2], which ends every other wait of the launch at once, and reports info = -7; launch_potrf turns that into TGP_RC_HANDOFF
// and the solve entry points run the solve again without this kernel (api.hip: gp_solve_once).
constexpr unsigned long long PANEL_SPIN_TICKS = 5000000ull;
constexpr int PANEL_SYNC_WORDS = 16;           // 64 B per panel: [0] A1 slices done, [1] A2 slices done, [2] abort
__device__ __forceinline__ void panel_publish(unsigned *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ bool panel_wait(unsigned *sync, int which, unsigned target, int *info) {
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        int ok = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        unsigned it = 0;
        while (__hip_atomic_load(sync + which, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++it & 63u) == 0u) {
                const bool late = __builtin_amdgcn_s_memrealtime() - t0 > PANEL_SPIN_TICKS;
                if (late || __hip_atomic_load(sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                    __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (late) __hip_atomic_store(info, -7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_ok = ok;
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    __syncthreads();                             // s_ok is read by everyone before a second wait overwrites it
    return ok;
}
template <bool SMALLROWS>
__global__ __launch_bounds__(256) void panel_mid_kernel(double *Pk, const double *W0, double *W1, int *info, int base, unsigned *sync) {
    // one LDS image for every role: potrf128's 96 KB; the 128-row GEMM tile stages its operands in the first 70 KB of it
    __shared__ __attribute__((aligned(16))) double T[potrf_v2::POTRF_LDS_DOUBLES];
    static_assert(potrf_v2::POTRF_LDS_DOUBLES >= 2 * 2 * 128 * TileDefault::LS, "the GEMM tile's staging fits in potrf128's image");
    static_assert(potrf_v2::POTRF_LDS_DOUBLES >= NT_SLICE_LDS_DOUBLES, "... and the latency tile's");
    TGP_CHAIN_PRIO();
    const int b = blockIdx.x;
    double *R1 = Pk + (int64_t)TGP_TB * TGP_PW;                 // row 128 of the panel
    TGP_MID_STAMP(0);
    if (b < 8) {            // (whole slices: X overwrites the rows it is computed from, a column half would race with its sibling's loads)
        double *rows = R1 + (int64_t)b * 16 * TGP_PW;
        nt_slice_tile<0, TGP_TB, 1>(T, rows, TGP_PW, W0, TGP_TB, rows, TGP_PW, nullptr, nullptr);
        TGP_MID_STAMP(1);
        panel_publish(sync + 0);
        TGP_MID_STAMP(2);
        return;
    }
    if (b < 24) {           // slice (b - 8) >> 1, column half b & 1
        if (!panel_wait(sync, 0, 8u, info)) return;
        TGP_MID_STAMP(1);
        double *rows = R1 + (int64_t)((b - 8) >> 1) * 16 * TGP_PW;
        const int64_t ch = (int64_t)(b & 1) * 64;
        nt_slice_tile<1, TGP_TB, 1, 64>(T, rows, TGP_PW, R1 + ch * TGP_PW, TGP_PW, rows + TGP_TB + ch, TGP_PW, nullptr, nullptr);
        TGP_MID_STAMP(2);
        panel_publish(sync + 1);
        TGP_MID_STAMP(3);
        return;
    }
    if (b == 24) {
        if (!panel_wait(sync, 1, 16u, info)) return;
        TGP_MID_STAMP(1);
        potrf_v2::potrf128_body<true>(T, R1 + TGP_TB, TGP_PW, W1, info, base + TGP_TB);
        TGP_MID_STAMP(2);
        return;
    }
    if constexpr (SMALLROWS) {
        double *rows = Pk + (int64_t)TGP_PW * TGP_PW + (int64_t)(b - 25) * 16 * TGP_PW;
        nt_slice_tile<0, TGP_TB, 1>(T, rows, TGP_PW, W0, TGP_TB, rows, TGP_PW, nullptr, nullptr);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's X0 is in the L2 before the others read it
        __syncthreads();
        if (!panel_wait(sync, 0, 8u, info)) return;
        nt_slice_tile<1, TGP_TB, 1>(T, rows, TGP_PW, R1, TGP_PW, rows + TGP_TB, TGP_PW, nullptr, nullptr);
    } else {
        double *rows = Pk + (int64_t)TGP_PW * TGP_PW + (int64_t)(b - 25) * 128 * TGP_PW;
        gemm_tile_128_at<0, TGP_TB, TGP_TB>(T, rows, W0, rows);
        __syncthreads();
        if (!panel_wait(sync, 0, 8u, info)) return;
        gemm_tile_128_at<1, TGP_PW, TGP_TB>(T, rows, R1, rows + TGP_TB);
    }
}

template <int NSEG>
__global__ __launch_bounds__(256) void syrk_small_kernel(double *Abase, int64_t Np, int ob, int T, const double *P0,
                                                         const double *P1) {
    const int ri = blockIdx.x;                   // 16-row slice of the trailing matrix
    const int tj = blockIdx.y;                   // 128-column tile
    if (tj > (ri >> 3) || (ri >> 3) >= T) return;
    TGP_CHAIN_PRIO();
    const int64_t pj = ob + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * ob + 16 * (int64_t)ri;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    const int64_t oa = (int64_t)ri * 16 * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
    nt_slice_tile<1, TGP_PW, NSEG>(nt_slice_lds_storage(), P0 + oa, TGP_PW, P0 + obb, TGP_PW, C, TGP_PW, NSEG > 1 ? P1 + oa : nullptr,
                                   NSEG > 1 ? P1 + obb : nullptr);
}

// Trailing update C(ti, tj) -= sum over NSEG panels of P[ti] P[tj]^T on the lower-triangular tile set of the trailing
// matrix that starts at block `ob` (256-row blocks), on the DTV tile (gemm_tile.h), 128 x 128 per workgroup, 2 per CU.
// P0 / P1 point at the row of the factored panel(s) that corresponds to the first trailing row.
//   strip == 0 : all tiles tj <= ti < T, XCD-aware super-tile enumeration (tilemap)
//   strip  > 0 : only the first `strip` tile columns (the columns the next panels live in), so
//                that their factorisation can start before the rest of the update has finished
template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_dtv_kernel(double *Abase, int64_t Np, int ob, int T, int strip, const double *P0,
                                                          const double *P1) {
    int ti, tj;
    if (strip == 0) {
        tilemap(blockIdx.x, T, ti, tj);
        if (ti < 0) return;
    } else {
        tj = (int)(blockIdx.x % strip);
        ti = (int)(blockIdx.x / strip);
        if (ti < tj || ti >= T) return;
    }
    if (strip) TGP_CHAIN_PRIO();
    const int64_t pj = ob + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * ob + (int64_t)TGP_TB * ti;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    const int64_t oa = (int64_t)ti * TGP_TB * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
    gemm_tile_dtv<4, TGP_PW, NSEG>(P0 + oa, P0 + obb, C, NSEG > 1 ? P1 + oa : nullptr, NSEG > 1 ? P1 + obb : nullptr);
}

// Strip updates on the chain's critical path (the tile columns of the next panels) with 64-row tiles: each wave owns
// 16 x 128 of C instead of 32 x 128, so a tile takes half as long, and while the strip fits the chip in one round either
// way (twice the workgroups, still at most two rounds up to T = 128) the panel chain waits about half as long for it.
template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_strip64_kernel(double *Abase, int64_t Np, int ob, int T, int strip, const double *P0,
                                                              const double *P1, int ti_min = 0) {
    const int tj = (int)(blockIdx.x % strip);        // 128-column tile
    const int th = (int)(blockIdx.x / strip) + 2 * ti_min;        // 64-row half tile (ti_min: skip the first tile rows -- the
    const int ti = th >> 1;                                       // diagonal block somebody else updates)
    if (ti < tj || ti >= T) return;
    TGP_CHAIN_PRIO();
    const int64_t pj = ob + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * ob + (int64_t)64 * th;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    const int64_t oa = (int64_t)th * 64 * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
    gemm_tile_dtv<4, TGP_PW, NSEG, 1>(P0 + oa, P0 + obb, C, NSEG > 1 ? P1 + oa : nullptr, NSEG > 1 ? P1 + obb : nullptr);
}

// ... and with 32-row tiles (gemm_tile_dtv32) where twice as many workgroups again still find the chip in one round
template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_strip32_kernel(double *Abase, int64_t Np, int ob, int T, int strip, const double *P0,
                                                              const double *P1) {
    const int tj = (int)(blockIdx.x % strip);        // 128-column tile
    const int tq = (int)(blockIdx.x / strip);        // 32-row quarter tile
    const int ti = tq >> 2;
    if (ti < tj || ti >= T) return;
    TGP_CHAIN_PRIO();
    const int64_t pj = ob + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * ob + (int64_t)32 * tq;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    const int64_t oa = (int64_t)tq * 32 * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
    gemm_tile_dtv32<TGP_PW, NSEG>(P0 + oa, P0 + obb, C, NSEG > 1 ? P1 + oa : nullptr, NSEG > 1 ? P1 + obb : nullptr);
}

// The depth-512 bulk update as a PERSISTENT grid that leaves part of the chip free, for steps where the serial panel
// chain (side stream) is longer than the update.  A bulk workgroup lives ~126 us and a plain launch fills both
// workgroup slots of all 256 CUs at once, so the chain's kernels -- potrf128 needs a CU with a free slot, the panel
// GEMMs one slot per 128 rows -- wait for the first round of tiles to finish (measured at N = 8192: 240 us of a 660 us
// cycle).  Here only gridDim.x < 512 workgroups exist and they take tiles from a queue: the slots they do not occupy
// stay free for the chain.  One counter per XCD class keeps the XCD-aware tile map (blockIdx.x & 7 = XCD of the
// workgroup); the loop ends for every workgroup once its class has run out of slots.
template <int NSEG>
// `half_from`: tile slots of a class from this one on are taken as two 64 x 128 half tiles each (two queue entries): the launch
// ends when its slowest workgroup does, on average half a tile time after the queues run dry (63 us of a 570 us launch at
// N = 8192), and half tiles in the last round halve that.  `entries` = half_from + 2 (slots - half_from) queue entries per class.
__global__ __launch_bounds__(256, 2) void syrk_dtv_queue_kernel(double *Abase, int64_t Np, int ob, int T, unsigned entries, unsigned half_from, int nres,
                                                                unsigned *__restrict__ queue, const double *P0, const double *P1, int steal) {
    __shared__ unsigned s_slot;
    const unsigned xcd = blockIdx.x & 7;
    {   // `nres` compute units per shader engine and XCD (32 nres of 256) are kept clear of this kernel: the first
        // workgroups to arrive on a shader engine name their own CUs (queue[8 + 3 (4 XCC_ID + SE_ID) + r]) and leave, and
        // so does every later one that lands there.  Per shader engine, because the dispatcher deals the workgroups of a
        // kernel to the shader engines in strict rotation and stalls on a full one: free CUs on a single engine admit one
        // or two workgroups of the chain's kernels and the rest wait behind them (measured with in-kernel stamps,
        // tools/potrf_stamps.py).
        if (threadIdx.x == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));      // HW_ID[15:8]: CU_ID[3:0], SH_ID, SE_ID[2:0]
            const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;  // XCC_ID[3:0]
            const unsigned key = hw + 1;
            unsigned *claims = queue + 8 + TGP_QUEUE_MAXRES * (4 * xcc + ((hw >> 5) & 3));
            unsigned leave = 0;
            for (int r = 0; r < nres && !leave; ++r) {
                const unsigned prev = atomicCAS(&claims[r], 0u, key);
                leave = (prev == 0u || prev == key) ? 1u : 0u;
            }
            // Progress does not depend on where the dispatcher puts workgroups: at most 8 nres + 8 of the 65 of a tile
            // class may leave.  (With other contexts' kernels on the chip the only free room can be the units those kernels
            // keep clear; without the cap every workgroup of this one could land there and leave, and no tile be done.)
            if (leave && atomicAdd(&queue[TGP_QUEUE_LEAVE + xcd], 1u) >= 8u * (unsigned)nres + 8u) leave = 0u;
            s_slot = leave;
        }
        __syncthreads();
        const bool leave = s_slot != 0u;
        __syncthreads();
#ifdef TGP_POTRF_STAMPS
        if (threadIdx.x == 0 && T == tgp_queue_stamp_T) {
            tgp_queue_stamps[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_memrealtime();
            tgp_queue_stamps[blockIdx.x * 4 + 2] = (leave ? 1ull << 32 : 0ull) | (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 8) |
                                                   __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
            tgp_queue_stamps[blockIdx.x * 4 + 1] = tgp_queue_stamps[blockIdx.x * 4 + 0];
            tgp_queue_stamps[blockIdx.x * 4 + 3] = 0;
        }
#endif
        if (leave) return;
    }
#ifdef TGP_POTRF_STAMPS
    unsigned my_tiles = 0;
#endif
    // A class that has run dry helps the next one (TGP_QUEUE_STEAL, default on): the classes hold 248 - 266 tiles at T = 56 (diagonal
    // super-tiles are not full) and the launch used to end 50 us after its average workgroup (in-kernel stamps, N = 8192).
    unsigned cls = xcd;
    int dry = 0;
    for (;;) {
        if (threadIdx.x == 0) s_slot = atomicAdd(&queue[cls], 1u);
        __syncthreads();
        unsigned n = s_slot;
        __syncthreads();                         // s_slot has been read by everyone before the next round overwrites it
        if (n >= entries) {                      // uniform
            if (!steal || ++dry == 8) break;
            cls = (cls + 1u) & 7u;
            continue;
        }
        int half = -1;
        if (n >= half_from) {
            half = (int)((n - half_from) & 1u);
            n = half_from + ((n - half_from) >> 1);
        }
        int ti, tj;
        tilemap(((int64_t)n << 3) | cls, T, ti, tj);
        if (ti < 0) continue;
        const int64_t pj = ob + (tj >> 1);
        const int64_t I = (int64_t)TGP_PW * ob + (int64_t)TGP_TB * ti;
        double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
        const int64_t oa = (int64_t)ti * TGP_TB * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
        if (half < 0) {
            gemm_tile_dtv<4, TGP_PW, NSEG>(P0 + oa, P0 + obb, C, NSEG > 1 ? P1 + oa : nullptr, NSEG > 1 ? P1 + obb : nullptr);
        } else {
            const int64_t ho = (int64_t)half * 64 * TGP_PW;
            gemm_tile_dtv<4, TGP_PW, NSEG, 1>(P0 + oa + ho, P0 + obb, C + ho, NSEG > 1 ? P1 + oa + ho : nullptr, NSEG > 1 ? P1 + obb : nullptr);
        }
#ifdef TGP_POTRF_STAMPS
        if (threadIdx.x == 0 && T == tgp_queue_stamp_T) {
            tgp_queue_stamps[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
            tgp_queue_stamps[blockIdx.x * 4 + 3] = ++my_tiles;
        }
#endif
    }
}

// "The head of a fused launch is done" (multi-GPU bulk update): the first `nhead` workgroup indices of a launch are the tile columns the panel chain
// waits for; each of them counts itself in when its tile is stored (valid or not: the count is the grid's), and the last one
// publishes the sequence number on which the chain's stream is parked (hipStreamWaitValue32).  One launch then carries the
// head AND the rest of the update: no second ramp and tail.
struct HeadSignal {
    unsigned *done = nullptr, *flag = nullptr;      // counter (monotonic over launches, modulo 2^32) / where to publish
    unsigned target = 0, seq = 0;                   // value of *done that completes this launch's head / what to publish then
    int nhead = 0;                                  // 0: not a fused launch
};
__device__ __forceinline__ void head_done(const HeadSignal &h) {
    // The workgroup's stores become visible device-wide: every wave waits until its own stores have been acknowledged by the
    // L2 (vmcnt(0)), the barrier collects the waves, and ONE wave then writes the XCD's L2 back (release fence at agent scope);
    // then one thread counts the workgroup in.  (Every wave fencing before the barrier -- four write-backs per workgroup --
    // cost 1.5 ms of a rank's 190 at 8 ranks and N = 65 536: profiles/r04_head_fence_ab.txt.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            // acquire + release: the workgroup that sees the target value has then acquired every other head workgroup's
            // release, so the flag store below orders ALL head tiles by the memory model, not by fence placement
            const unsigned old = __hip_atomic_fetch_add(h.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1u == h.target) __hip_atomic_store(h.flag, h.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// Trailing update with a compile-time list of NSEG factored panels (depth 256 * NSEG): P.a[s] points at the row of
// panel s that corresponds to the first trailing row (P.b is filled in per tile).  Same tile map as above.
// (The fused head + rest form of the multi-GPU kernel below was measured here too in round 4 and is NOT used: 1368.7 vs
// 1361.0 ms at N = 65 536, profiles/r04_fused_bulk_ab.txt -- the rest tiles start in lockstep when the head ends, so the
// panel chain's first kernel waits a whole tile time for a slot, where after a separate head launch it finds the chip empty.)
template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_segs_kernel(double *Abase, int64_t Np, int ob, int T, int strip, SegPtrs<NSEG> P) {
    int ti, tj;
    if (strip == 0) {
        tilemap(blockIdx.x, T, ti, tj);
        if (ti < 0) return;
    } else {
        tj = (int)(blockIdx.x % strip);
        ti = (int)(blockIdx.x / strip);
        if (ti < tj || ti >= T) return;
    }
    if (strip) TGP_CHAIN_PRIO();
    const int64_t pj = ob + (tj >> 1);
    const int64_t I = (int64_t)TGP_PW * ob + (int64_t)TGP_TB * ti;
    double *C = Abase + panel_off(pj, Np) + (I - pj * TGP_PW) * TGP_PW + (tj & 1) * TGP_TB;
    const int64_t oa = (int64_t)ti * TGP_TB * TGP_PW, obb = (int64_t)tj * TGP_TB * TGP_PW;
    SegPtrs<NSEG> sp;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        sp.a[s] = P.a[s] + oa;
        sp.b[s] = P.a[s] + obb;
    }
    gemm_tile_dtv_segs<4, TGP_PW, NSEG>(sp, C);
}

// multi-GPU trailing update: rank g updates its own block rows, after a GROUP of NSEG consecutive panels
// kpanel .. kpanel+NSEG-1, in one pass of depth 256 NSEG.
// Gathered panel s holds the blocks > kpanel+s ([rank][cmax[s]][256][256]); tiles are counted from block kpanel+NSEG.
template <int NSEG>
struct DistSegs {
    const double *P[NSEG];    // gathered panel s: where the ROW operand (this rank's rows) is read from
    const double *PB[NSEG];   // ... and the COLUMN operand (the rows of the tile column's block); == P[s] for a gathered panel
    int cmax[NSEG];
    int frg[NSEG];            // dist_first_round(kpanel + s + 1, g, G): this rank's first local index inside gathered panel s
};
// Tile enumeration of a rank's share, XCD-aware like the single-GPU tilemap: blocks b, b+8, ... share an XCD (and its
// L2); each XCD works through 8 x 8 super-tiles (8 local tile rows against 8 tile columns) dealt round-robin.  The
// share is a staircase (local tile row lt reaches up to its own global column), so only super-tiles that contain
// valid tiles are enumerated: `start` holds, per group of 8 local tile rows, the index of its first super-tile.
// A rank's launches are short (2 .. 30 rounds of 512 tiles at 8 ranks and N = 65 536), so two things that do not matter
// on one GPU do here:
//  * whole super-tiles dealt in turn leave the XCDs up to one super-tile = one full round of an XCD's 64 slots apart:
//    only the first 8 floor(total / 8) super-tiles are dealt whole, every one of the remaining (< 8) is cut into its 8
//    tile rows, one per XCD (`whole`);
//  * the launch ends when its slowest workgroup does, on average half a tile time (125 us at depth 1024) after the slots
//    start to idle: the last round's tiles (`half_from`) are run as two 64 x 128 half tiles each, which halves that;
//  * the tile columns of the NEXT group of panels (`head_cols`, which the panel chain waits for) and the rest used to
//    be two launches, each with its own ramp and tail: in the fused form the first `nhead` workgroup indices are the
//    dense grid of the head columns, the super-tile map of the rest follows, and the workgroup that completes the
//    last head tile publishes `seq` to `flag`, on which the chain's stream is parked (hipStreamWaitValue32).
struct DistMap {
    int start[258];           // prefix of super-tiles per row group; start[ngroups] = total  (N up to 262144 on one rank)
    int ngroups;              // 0: dense enumeration only (strips: workgroup b = local tile row b / ncol, column b % ncol)
    int qfb;                  // local index (= round) of the first block >= kpanel + NSEG this rank owns (host: dist_first_round)
    unsigned ginv;            // floor(2^32 / G) + 1: x / G == (x * ginv) >> 32 for 0 <= x < 65536 (block indices are < 1024); 0 for G == 1
    int whole;                // 64 floor(total / 8): slots per XCD class that belong to whole super-tiles
    int half_from;            // tile slots of a class from this one on are run as two 64-row half tiles each (the last round)
    int nhead, head_cols;     // fused form: workgroup indices below nhead (a multiple of 8) are the dense grid of the head columns
    int head_half;            // the dense grid (a strip, or the head of a fused launch) runs as 64-row half tiles, two workgroups per tile
    HeadSignal hs;            // what the head's last workgroup publishes (hs.nhead is not used here)
};
__device__ __forceinline__ int div_g(int x, unsigned ginv) { return ginv ? (int)__umulhi((unsigned)x, ginv) : x; }     // ginv == 0: G == 1
// the reflected owner map of tgp_internal.h (dist_owner / dist_block_of / dist_first_round) in 32 bits: position of rank r in
// round q and, the same involution, rank at position r of round q
__device__ __forceinline__ int snake_pos(int q, int r, int G) { return (q & 1) ? G - 1 - r : r; }
__device__ __forceinline__ int first_round32(int s, int r, int G, unsigned ginv) {
    const int q = div_g(s, ginv);
    return q + (snake_pos(q, r, G) < s - G * q ? 1 : 0);
}
// one tile of the rank's share: b is the (virtual) workgroup index of the plain launch.  Index arithmetic in 32 bits with a
// reciprocal of G from the host -- the run-time-G 64-bit divisions of round 2's version (a dozen per tile, scalar code)
// were 1.6 % of the kernel at depth 1024 (world of one, A/B against the single-GPU kernel inside this driver).
// Block b' of rank r sits at index b' / G - dist_first_round(first, r, G) among r's blocks >= first.
// Every thread of the workgroup takes the same path (b is uniform).
template <int NSEG>
__device__ __forceinline__ void syrk_distn_tile(int64_t b, double *Aloc, const int64_t *__restrict__ loff, int kpanel, int G, int g,
                                                const DistSegs<NSEG> &S, int col_lo, int ncol, int nrows, const DistMap &M) {
    int lt, ct, half = -1;
    bool valid = true;
    const bool head = M.ngroups == 0 || b < M.nhead;
    if (head) {
        const int hc = M.ngroups == 0 ? ncol : M.head_cols;
        int64_t bb = b;
        if (M.head_half) {                    // what the panel chain waits for, when it is less than a round of tiles: half the time per tile
            half = (int)(bb & 1);
            bb >>= 1;
        }
        lt = (int)(bb / hc);
        ct = (int)(bb - (int64_t)lt * hc);
        valid = lt < nrows;
    } else {
        const int bb = (int)(b - M.nhead);
        const int x = bb & 7;
        int n = bb >> 3;
        if (n >= M.half_from) {               // two consecutive slots share a tile: rows 0..63 and 64..127
            half = (n - M.half_from) & 1;
            n = M.half_from + ((n - M.half_from) >> 1);
        }
        int st, within;
        if (n < M.whole) {
            st = (n >> 6) * 8 + x;
            within = n & 63;
        } else {                              // the last (< 8) super-tiles: tile row x of each goes to XCD class x
            const int nn = n - M.whole;
            st = (M.whole >> 3) + (nn >> 3);
            within = x * 8 + (nn & 7);
        }
        valid = st < M.start[M.ngroups];
        int R = 0;
        for (int step = 128; step > 0; step >>= 1)
            if (R + step <= M.ngroups && M.start[R + step] <= st) R += step;      // last group with start <= st
        lt = R * 8 + (within >> 3);
        ct = M.head_cols + (st - M.start[R]) * 8 + (within & 7);
        valid = valid && lt < nrows && ct < ncol;
    }
    const int gtj = ct + col_lo;
    const int s0 = kpanel + NSEG;
    const int qi = M.qfb + (lt >> 1);
    const int bi = qi * G + snake_pos(qi, g, G);
    const int gti = 2 * (bi - s0) + (lt & 1);
    valid = valid && gtj <= gti;
    if (valid) {
        if (head) TGP_CHAIN_PRIO();           // the columns the panel chain waits for (strips, or the head of a fused launch)
        const int bj = s0 + (gtj >> 1);
        const int qj = div_g(bj, M.ginv);
        const int rj = snake_pos(qj, bj - G * qj, G);
        const int64_t hi = (lt & 1) * TGP_TB, hj = (gtj & 1) * TGP_TB;
        SegPtrs<NSEG> sp;
#pragma unroll
        for (int s = 0; s < NSEG; ++s) {
            const int first = kpanel + s + 1;                     // first block held by gathered panel s
            const int is = qi - S.frg[s], js = qj - first_round32(first, rj, G, M.ginv);
            sp.a[s] = S.P[s] + (((int64_t)g * S.cmax[s] + is) * TGP_PW + hi) * TGP_PW;
            sp.b[s] = S.PB[s] + (((int64_t)rj * S.cmax[s] + js) * TGP_PW + hj) * TGP_PW;
        }
        double *c = Aloc + loff[bj] + ((int64_t)(qi - first_round32(bj, g, G, M.ginv)) * TGP_PW + hi) * TGP_PW + hj;
        if (half < 0) {
            gemm_tile_dtv_segs<4, TGP_PW, NSEG>(sp, c);
        } else {
#pragma unroll
            for (int s = 0; s < NSEG; ++s) sp.a[s] += (int64_t)half * 64 * TGP_PW;
            gemm_tile_dtv_segs<4, TGP_PW, NSEG, 1>(sp, c + (int64_t)half * 64 * TGP_PW);
        }
        if (head && M.ngroups != 0) __builtin_amdgcn_s_setprio(0);
    }
    if (M.ngroups != 0 && b < M.nhead) head_done(M.hs);       // head tile of a fused launch, valid or not
}

template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_distn_kernel(double *Aloc, const int64_t *__restrict__ loff, int kpanel, int G,
                                                            int g, DistSegs<NSEG> S, int col_lo, int ncol, int nrows,
                                                            DistMap M) {
    syrk_distn_tile<NSEG>(blockIdx.x, Aloc, loff, kpanel, G, g, S, col_lo, ncol, nrows, M);
}

// The same update as a PERSISTENT grid that keeps `nres` compute units per shader engine and XCD clear for the panel chain
// (see syrk_dtv_queue_kernel: what the chain waits for beside a plain bulk launch, and why the clear units are per shader
// engine): for the steps of the multi-GPU factorisation whose local share of the bulk is shorter than the panel chain --
// every step of the second half at 8 ranks and N = 65 536 -- where a diagonal block next to bulk waves takes 170 - 230 us
// instead of 44.  Workgroups take the plain launch's workgroup indices from per-XCD-class queues.
template <int NSEG>
__global__ __launch_bounds__(256, 2) void syrk_distn_queue_kernel(double *Aloc, const int64_t *__restrict__ loff, int kpanel, int G,
                                                                  int g, DistSegs<NSEG> S, int col_lo, int ncol, int nrows, DistMap M,
                                                                  unsigned slots_per_class, int nres, unsigned *__restrict__ queue) {
    __shared__ unsigned s_slot;
    const unsigned xcd = blockIdx.x & 7;
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));      // HW_ID[15:8]: CU_ID[3:0], SH_ID, SE_ID[2:0]
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;  // XCC_ID[3:0]
        const unsigned key = hw + 1;
        unsigned *claims = queue + 8 + TGP_QUEUE_MAXRES * (4 * xcc + ((hw >> 5) & 3));
        unsigned leave = 0;
        for (int r = 0; r < nres && !leave; ++r) {
            const unsigned prev = atomicCAS(&claims[r], 0u, key);
            leave = (prev == 0u || prev == key) ? 1u : 0u;
        }
        // at most 8 nres + 8 workgroups of a class may leave: progress never depends on where the dispatcher puts workgroups
        if (leave && atomicAdd(&queue[TGP_QUEUE_LEAVE + xcd], 1u) >= 8u * (unsigned)nres + 8u) leave = 0u;
        s_slot = leave;
    }
    __syncthreads();
    const bool leave = s_slot != 0u;
    __syncthreads();
    if (leave) return;
    for (;;) {
        if (threadIdx.x == 0) s_slot = atomicAdd(&queue[xcd], 1u);
        __syncthreads();
        const unsigned n = s_slot;
        __syncthreads();                         // s_slot has been read by everyone before the next round overwrites it
        if (n >= slots_per_class) break;         // uniform
        syrk_distn_tile<NSEG>(((int64_t)n << 3) | xcd, Aloc, loff, kpanel, G, g, S, col_lo, ncol, nrows, M);
        __syncthreads();                         // the tile's LDS staging is done before the next tile reuses it
    }
}

// `exclusive`: ask for so much LDS (128 KB in all) that no trailing-update workgroup fits on the compute unit beside this
// one.  A bulk wave on the same SIMD issues fp64 MFMAs back to back and each of them holds the fp64 pipe for 64 cycles:
// every DEPENDENT instruction of the factorisation's serial chain then waits for one (in-kernel stamps at N = 8192: the
// 32x32 diagonal steps 14k -> 72-88k cycles, the whole block 44 -> 156 us).  Only used where free compute units are
// guaranteed (the queued bulk update below keeps one per XCD clear); elsewhere the kernel would wait for a CU to drain.
// `solo`: the launch has a compute unit to itself (nothing else on the chip, or `exclusive`): the variant without the register cap
inline void run_potrf128(hipStream_t st, double *A, int lda, double *W, int *info, int base, bool exclusive = false, bool solo = false) {
    constexpr unsigned image = potrf_v2::POTRF_LDS_BYTES, full = 128 * 1024;
    static const bool attr_ok = [] {
        return hipFuncSetAttribute((const void *)potrf_v2::potrf128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)full) == hipSuccess &&
               hipFuncSetAttribute((const void *)potrf_v2::potrf128_solo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)full) == hipSuccess;
    }();
    (void)attr_ok;          // (if the attribute could not be set the launch below fails and launch_potrf reports it)
    if (exclusive || solo) potrf_v2::potrf128_solo_kernel<<<1, 256, exclusive ? full : image, st>>>(A, lda, W, info, base);
    else potrf_v2::potrf128_kernel<true><<<1, 256, image, st>>>(A, lda, W, info, base);
}
}  // namespace

namespace {
// factor one 256-wide panel: the two 128x128 diagonal blocks (L + inverse) and the rows below them
__global__ void set_identity128_kernel(double *__restrict__ W) {
    const int t = blockIdx.x * 256 + threadIdx.x;          // 64 x 256 threads: one element each
    W[t] = (t >> 7) == (t & 127) ? 1.0 : 0.0;
}

// `n_data` (>= 0): order of the matrix before padding.  When the second 128-block of a panel lies entirely in the padding
// (possible for the last panel only) it is the identity and its own factor: no potrf128, no update of it, W1 = I.
// `mid_sync` != nullptr: the step between the two diagonal blocks runs as panel_mid_kernel (one launch, in-kernel hand-offs, the
// second diagonal block inside it); `cu_budget` = compute units its workgroups can expect to find free (each takes a whole one)
void factor_panel(hipStream_t st, double *Pk, int64_t mk, double *W0, int *d_info, int base, bool exclusive = false,
                  int64_t n_data = -1, unsigned *mid_sync = nullptr, int cu_budget = 0) {
    double *W1 = W0 + TGP_TB * TGP_TB;
    double *R1 = Pk + (int64_t)TGP_TB * TGP_PW;                 // row 128 of the panel
    const int r1 = (int)((mk - TGP_TB) / TGP_TB);
    // few row blocks: 16-row slices spread a block over 8 workgroups (latency); many: 128-row tiles (throughput)
    static const int small_rows = [] { const char *e = getenv("TGP_SMALL_ROWS"); return e ? atoi(e) : 80; }();      // (40 until the latency tile was staged through LDS; 24 / 40 / 56 / 64 / 80 / 96: profiles/r05_slice_tile_ab.txt)
    const bool solo = cu_budget >= 256 && mid_sync != nullptr;      // callers pass the whole chip only where the chain runs alone
    run_potrf128(st, Pk, TGP_PW, W0, d_info, base, exclusive, solo);
    if (n_data >= 0 && (int64_t)base + TGP_TB >= n_data && mk == TGP_PW) {
        gemm_col_small_kernel<0, TGP_TB><<<8, 256, 0, st>>>(R1, W0, R1);      // rows 128..255: zero, or the right-hand side row
        set_identity128_kernel<<<64, 256, 0, st>>>(W1);
        return;
    }
    const int r2 = (int)((mk - TGP_PW) / TGP_TB);
    if (mid_sync) {
        constexpr unsigned pad = 128 * 1024 - potrf_v2::POTRF_LDS_DOUBLES * 8;      // exclusive: a compute unit per workgroup
        static const bool pad_ok = [] {
            return hipFuncSetAttribute((const void *)panel_mid_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad) == hipSuccess &&
                   hipFuncSetAttribute((const void *)panel_mid_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad) == hipSuccess;
        }();
        const unsigned dyn = (exclusive && pad_ok) ? pad : 0u;
        if (17 + 8 * r2 <= cu_budget) panel_mid_kernel<true><<<25 + 8 * r2, 256, dyn, st>>>(Pk, W0, W1, d_info, base, mid_sync);
        else panel_mid_kernel<false><<<25 + r2, 256, dyn, st>>>(Pk, W0, W1, d_info, base, mid_sync);
    } else if (cu_budget > 0) {
        // A step that WOULD run as panel_mid_kernel, were this solve alone on the chip: the same arithmetic as separate launches --
        // rows 128..255 as eight 16-row slices, the rows below in the tile form that kernel would choose -- so that a solve's
        // bits do not depend on whether another context happened to be busy (the two tile forms sum in different orders).
        double *R2 = Pk + (int64_t)TGP_PW * TGP_PW;
        const bool small = 17 + 8 * r2 <= cu_budget;
        gemm_col_small_kernel<0, TGP_TB><<<8, 256, 0, st>>>(R1, W0, R1);
        if (r2 > 0) {
            if (small) gemm_col_small_kernel<0, TGP_TB><<<r2 * 8, 256, 0, st>>>(R2, W0, R2);
            else gemm_col_kernel<0, TGP_TB><<<r2, 256, 0, st>>>(R2, W0, R2);
        }
        gemm_col_small_kernel<1, TGP_PW><<<8, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
        if (r2 > 0) {
            if (small) gemm_col_small_kernel<1, TGP_PW><<<r2 * 8, 256, 0, st>>>(R2, R1, R2 + TGP_TB);
            else gemm_col_kernel<1, TGP_PW><<<r2, 256, 0, st>>>(R2, R1, R2 + TGP_TB);
        }
        run_potrf128(st, R1 + TGP_TB, TGP_PW, W1, d_info, base + TGP_TB, exclusive, solo);
    } else {
        if (r1 <= small_rows) {
            gemm_col_small_kernel<0, TGP_TB><<<r1 * 8, 256, 0, st>>>(R1, W0, R1);
            gemm_col_small_kernel<1, TGP_PW><<<r1 * 8, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
        } else {
            gemm_col_kernel<0, TGP_TB><<<r1, 256, 0, st>>>(R1, W0, R1);
            gemm_col_kernel<1, TGP_PW><<<r1, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
        }
        run_potrf128(st, R1 + TGP_TB, TGP_PW, W1, d_info, base + TGP_TB, exclusive, solo);
    }
    if (r2 > 0) {
        double *R2 = Pk + (int64_t)TGP_PW * TGP_PW + TGP_TB;    // row 256, column 128
        if (r2 <= small_rows) gemm_col_small_kernel<0, TGP_TB><<<r2 * 8, 256, 0, st>>>(R2, W1, R2);
        else gemm_col_kernel<0, TGP_TB><<<r2, 256, 0, st>>>(R2, W1, R2);
    }
}

inline int small_t() {          // steps with at most this many tile rows run on the latency tile (16-row slices)
    static const int v = [] { const char *e = getenv("TGP_SMALL_T"); return e ? atoi(e) : 8; }();
    return v;
}

template <int NSEG>
void launch_syrk(hipStream_t st, double *d_A, int64_t Np, int ob, int T, int strip, const double *P0, const double *P1) {
    if (T <= 0) return;
    // strips the panel chain waits for also run as 16-row slices further up (TGP_STRIP_SMALL_T tile rows): since the latency tile is
    // staged through LDS a depth-256 slice takes 7.8 us where a 32-row tile of the strip takes 15
    static const int strip_small_t = [] { const char *e = getenv("TGP_STRIP_SMALL_T"); return e ? atoi(e) : 16; }();      // (8 / 16 / 24 / 32 / 48: profiles/r05_slice_tile_ab.txt)
    if (T <= small_t() || (strip > 0 && T <= strip_small_t)) {
        const int cols = strip == 0 ? T : (strip < T ? strip : T);
        syrk_small_kernel<NSEG><<<dim3((unsigned)(T * 8), (unsigned)cols), 256, 0, st>>>(d_A, Np, ob, T, P0, P1);
        return;
    }
    static const int strip64_t = [] { const char *e = getenv("TGP_STRIP64_T"); return e ? atoi(e) : 128; }();
    static const int strip32_t = [] { const char *e = getenv("TGP_STRIP32_T"); return e ? atoi(e) : 64; }();
    if (strip > 0 && T <= strip32_t) {
        syrk_strip32_kernel<NSEG><<<(unsigned)((int64_t)4 * T * strip), 256, 0, st>>>(d_A, Np, ob, T, strip, P0, P1);
        return;
    }
    if (strip > 0 && T <= strip64_t) {
        syrk_strip64_kernel<NSEG><<<(unsigned)((int64_t)2 * T * strip), 256, 0, st>>>(d_A, Np, ob, T, strip, P0, P1);
        return;
    }
    const unsigned gs = strip == 0 ? (unsigned)tilemap_grid(T) : (unsigned)((int64_t)T * strip);
    syrk_dtv_kernel<NSEG><<<gs, 256, 0, st>>>(d_A, Np, ob, T, strip, P0, P1);
}

// The depth-512 bulk update after the pair (k, k+1) as the persistent grid that keeps compute units clear for the chain.
// How many of them (1 .. 3 per shader engine and XCD = 32 .. 96 CUs) by the trailing matrix's tile rows T, from the chain's own
// timeline (tools/chain_gaps.py: in-kernel stamps of the diagonal blocks, no profiler) with 1, 2 and 3 forced, per pair of
// panels at N = 8192 (profiles/r05_queue_res_per_pair.txt; cycle = chain + what it waits for the bulk's stream):
//   T       56    52    48    44    40    36    32    28    24    20
//   1      613   557   482   449   419   368   316   271   268   235    us per cycle
//   2      670   601   526   453   388   333   284   270   253   238
//   3      749   676   569   500   434   369   293   260   243   235
// Until round 5 the rule was "as many as leave the bulk shorter than the chain's ~400 us" with one threshold for both steps
// (TGP_QUEUE_BULK_US): 3 from T = 43 down, where the chain then waited 150 - 220 us per pair for the bulk.
int queued_nres(int T) {
    static const int queue_res = [] { const char *e = getenv("TGP_QUEUE_RES"); return e ? atoi(e) : 0; }();
    if (queue_res > 0) return queue_res > TGP_QUEUE_MAXRES ? TGP_QUEUE_MAXRES : queue_res;
    static const int t1 = [] { const char *e = getenv("TGP_QUEUE_T1"); return e ? atoi(e) : 42; }();      // one from here up
    static const int t2 = [] { const char *e = getenv("TGP_QUEUE_T2"); return e ? atoi(e) : 26; }();      // two from here up (30, 26, 22: equal within the spread; profiles/r05_queue_rule_ab.txt)
    return T >= t1 ? 1 : (T >= t2 ? 2 : 3);
}
void launch_syrk2_queued(tgp_ctx *ctx, hipStream_t st, double *d_A, int64_t Np, int ob, int T, const double *P0, const double *P1,
                         int nqueue) {
    const int nres = queued_nres(T);
    // the last round of a class (as many slots as it has resident workgroups) runs as half tiles; TGP_QUEUE_HALF = slots per class, 0 off
    static const int half_env = [] { const char *e = getenv("TGP_QUEUE_HALF"); return e ? atoi(e) : -1; }();
    const unsigned slots = (unsigned)(tilemap_grid(T) / 8);
    unsigned nhalf = half_env >= 0 ? (unsigned)half_env : (unsigned)(512 - 64 * nres) / 8u;
    if (nhalf > slots) nhalf = slots;
    static const int steal = [] { const char *e = getenv("TGP_QUEUE_STEAL"); return e ? atoi(e) : 1; }();
    syrk_dtv_queue_kernel<2><<<512 + 8, 256, 0, st>>>(d_A, Np, ob, T, slots + nhalf, slots - nhalf, nres,
                                                     ctx->d_queue + TGP_QUEUE_WORDS * nqueue, P0, P1, steal);
}
}  // namespace

// Right-looking factorisation.  Panels are taken in groups so that the bulk of the trailing matrix is updated once
// per 512 (pairs) or 1024 (groups of four) columns: the per-tile fixed costs of the update (C read + write, pipeline
// fill) are 13 % of a depth-256 launch, 6.8 % at depth 512 and 3.5 % at depth 1024.
//   pairs:  F(k) -> U1: panel k+1 only, depth 256 -> F(k+1) -> U2: everything right of it, depth 512
//   fours:  F(k) -> S1 -> F(k+1) -> S2 -> F(k+2) -> S3 -> F(k+3) -> U4: everything right of it, depth 1024
//           (Sj: the two tile columns of panel k+j against the j panels before it, one launch of depth 256 j)
// Both with look-ahead: the update is split into the tile columns of the next group (U2a / U4a) and the rest (U2b / U4b),
// and the next group is factored on a high-priority side stream under the rest.  The streams hand over to each other by
// tgp_signal / tgp_await (handoff.hip: flags + stream wait-value, or events).
// History at N=65536 (same tile): 1694 ms one panel at a time, 1571 pairs, 1551 pairs + look-ahead; with the DTV tile
// 1423 ms pairs + look-ahead, 1362 ms fours + look-ahead (68.9 TF, 87.6 % of the fp64 MFMA peak).
// Schedules that were measured and dropped (diagonal-first, a right-hand side riding along, head start for the chain) are
// in LAB_NOTES.md Appendix A and the git history.
int launch_potrf(tgp_ctx *ctx, double *d_A, int64_t Np, double *d_W, bool defer_info, int64_t n_data) {
    TGP_ARG(Np > 0 && Np % TGP_PW == 0);
    hipStream_t st = ctx->stream;
    TGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
    const int nP = (int)(Np / TGP_PW);
    const bool prof = ctx->profiling != 0;
    if (prof) {
        while ((int)ctx->prof_events.size() < 2 * nP) {
            hipEvent_t e;
            TGP_HIP(hipEventCreate(&e));
            ctx->prof_events.push_back(e);
        }
    }
    // TGP_CHOL_MODE: 3 = groups of four panels (depth-1024 bulk update) with a pair-wise tail, 2 = pairs with look-ahead,
    // 1 = pairs without look-ahead, 0 = one panel at a time.  Default: 3 from N = 18432 on (crossover re-measured with the faster panel
    // chain of round 5: equal at 16384, 46.8 vs 47.5 ms at 20480, 76.7 vs 78.7 at 24576; it was 22528 before), else 2.
    static const int mode_env = [] { const char *e = getenv("TGP_CHOL_MODE"); return e ? atoi(e) : -1; }();
    // Up to N = 1280 there is no bulk update worth a second stream (with flag hand-offs the look-ahead pays from Np = 1536 on
    // -- 0.650 vs 0.657 ms there, 0.906 vs 0.935 ms at 2048; with events it lost up to 2048: 0.51 vs 0.56 ms at N = 1024).
    // Contexts that run side by side (the concurrent likelihood evaluations of the ML fit) ask for
    // one stream each: the runtime has 4 hardware queues, and with two streams per context three contexts already share
    // queues and serialise (6 contexts at N = 1024: 0.60 ms per evaluation with look-ahead, 0.22 ms without).
    const int mode = mode_env >= 0 ? mode_env : (!ctx->lookahead ? 1 : (Np >= 18432 ? 3 : (Np <= 1280 ? 1 : 2)));
    if (mode >= 2) {
        int rc = tgp_ensure_side_stream(ctx);
        if (rc) return rc;
    }
    double flops = 0.0;
    int nlaunch = 0;
    // the roofline accounting (timings 5..7) covers ONE kernel: the bulk update of the schedule in use; in mode 3 the
    // pair-wise tail launches are run but not counted
    bool counting = true;
    auto timed = [&](auto &&fn, double fl) -> int {
        if (!counting) {
            fn();
            return 0;
        }
        if (prof) TGP_HIP(hipEventRecord(ctx->prof_events[2 * nlaunch], st));
        fn();
        if (prof) TGP_HIP(hipEventRecord(ctx->prof_events[2 * nlaunch + 1], st));
        flops += fl;
        ++nlaunch;
        return 0;
    };
    auto Wk = [&](int k) { return d_W + (int64_t)(2 * k) * TGP_TB * TGP_TB; };
    auto panel = [&](int k) { return d_A + panel_off(k, Np); };
    // ---- pairs with look-ahead (the schedule of TGP_CHOL_MODE=2, also the tail of mode 3) --------------------------
    // The update after pair (k, k+1) is split into the 4 tile columns the NEXT pair lives in (U2a) and the rest
    // (U2b); the next pair is factored on a high-priority side stream while U2b keeps the chip busy.
    static const int queue_t = [] { const char *e = getenv("TGP_QUEUE_T"); return e ? atoi(e) : 64; }();
    int nqueue = 0;
    // TGP_PANEL_MID=0: the step between a panel's two diagonal blocks as separate launches everywhere (before round 5)
    static const bool mid_env = [] { const char *e = getenv("TGP_PANEL_MID"); return e ? atoi(e) != 0 : true; }();
    // ... and only for a solve that started alone and can be run again (tgp_internal.h: tgp_solves_in_flight; api.hip: gp_solve_once)
    const bool mid_on = mid_env && !ctx->mid_off && ctx->mid_allowed;
    if (mid_on) TGP_HIP(hipMemsetAsync(ctx->d_psync, 0, (size_t)(nP < TGP_PSYNC_PANELS ? nP : TGP_PSYNC_PANELS) * PANEL_SYNC_WORDS * sizeof(unsigned), st));
    if (Np / TGP_TB - 8 > small_t())          // some step can take the queued form (T3 = Np/128 - 8 at the first pair)
        TGP_HIP(hipMemsetAsync(ctx->d_queue, 0, TGP_NQUEUE * TGP_QUEUE_WORDS * sizeof(unsigned), st));
    auto psync = [&](int k) { return (mid_on && k < TGP_PSYNC_PANELS) ? ctx->d_psync + PANEL_SYNC_WORDS * k : nullptr; };
    auto run_pairs = [&](int kstart) -> int {
        hipStream_t sd = ctx->side_stream;
        // `cus` > 0: the chain has that many compute units to itself (the whole chip, or the ones a queued bulk update keeps
        // clear) -- the step between a panel's two diagonal blocks then runs as panel_mid_kernel
        // `tell`: say on flag 2 when panel k is complete (its part of the next U2a is then applied early, see below)
        auto factor_pair = [&](hipStream_t s, int k, bool exclusive = false, int cus = 0, bool tell = false) {       // F(k), U1(k), F(k+1)
            const int64_t mk = Np - (int64_t)TGP_PW * k;
            auto sync = [&](int kk) { return (cus > 0 && mid_on && kk < TGP_PSYNC_PANELS) ? ctx->d_psync + PANEL_SYNC_WORDS * kk : nullptr; };
            factor_panel(s, panel(k), mk, Wk(k), ctx->d_info, k * TGP_PW, exclusive, n_data, sync(k), cus);
            if (tell) (void)tgp_signal(ctx, s, 2, ctx->ev[6]);
            if (k + 1 >= nP) return;
            const int T1 = (int)((mk - TGP_PW) / TGP_TB);
            launch_syrk<1>(s, d_A, Np, k + 1, T1, 2, panel(k) + (int64_t)TGP_PW * TGP_PW, nullptr);
            factor_panel(s, panel(k + 1), mk - TGP_PW, Wk(k + 1), ctx->d_info, (k + 1) * TGP_PW, exclusive, n_data, sync(k + 1), cus);
        };
        factor_pair(st, kstart, false, 256);
        // Chain-bound steps (T3 <= split_t): U2a -- the 4 tile columns of the next pair, depth 512, between F(k+1) and F(k+2) on
        // the critical path -- is taken in two halves of depth 256: panel k's half as soon as panel k is complete, on the bulk's
        // stream beside U1 and F(k+1); only panel k+1's half is left between the pairs (a strip tile of depth 256 takes ~22 us,
        // one of depth 512 ~42).
        static const int split_t = [] { const char *e = getenv("TGP_U2A_SPLIT_T"); return e ? atoi(e) : 64; }();
        bool early = false;         // panel k's half of U2a(k) has been applied already
        for (int k = kstart; k + 2 < nP; k += 2) {
            const int T2 = (int)((Np - (int64_t)TGP_PW * (k + 2)) / TGP_TB);       // tiles from block k+2
            const double *P0 = panel(k) + (int64_t)2 * TGP_PW * TGP_PW;
            const double *P1 = panel(k + 1) + (int64_t)TGP_PW * TGP_PW;
            const int T3 = T2 - 4;
            // chain-bound steps: the bulk runs as a persistent grid that keeps one compute unit per shader engine clear
            // for the side stream, and the diagonal blocks insist on a compute unit of their own
            const bool queued = T3 <= queue_t && T3 > small_t() && nqueue < TGP_NQUEUE;
            {   // U2a: tile columns 0..3 (panels k+2, k+3)
                const double rows = (double)T2 * TGP_TB, w = (T2 < 4 ? T2 : 4) * (double)TGP_TB;
                const double elems = w * (rows - w) + w * (w + 1.0) / 2.0;
                int rc = timed([&] {
                    if (early) launch_syrk<1>(st, d_A, Np, k + 2, T2, 4, P1, nullptr);
                    else launch_syrk<2>(st, d_A, Np, k + 2, T2, 4, P0, P1);
                }, (early ? 1.0 : 2.0) * 2.0 * TGP_PW * elems);
                if (rc) return rc;
            }
            TGP_HIP(tgp_signal(ctx, st, 0, ctx->ev[4]));
            TGP_HIP(tgp_await(ctx, sd, 0, ctx->ev[4]));
            const bool early_next = T3 > 0 && T3 <= split_t;
            // (steps with at most small_t tile rows have no bulk update to speak of: the chain has the chip to itself)
            factor_pair(sd, k + 2, queued, queued ? 32 * queued_nres(T3) : (T3 <= small_t() ? 256 : 0), early_next);
            TGP_HIP(tgp_signal(ctx, sd, 1, ctx->ev[5]));
            if (T3 > 0) {   // U2b: everything from block k+4 on
                const double m = (double)T3 * TGP_TB;
                const int64_t skip = (int64_t)4 * TGP_TB * TGP_PW;
                int rc = timed([&] {
                    if (queued) launch_syrk2_queued(ctx, st, d_A, Np, k + 4, T3, P0 + skip, P1 + skip, nqueue++);
                    else launch_syrk<2>(st, d_A, Np, k + 4, T3, 0, P0 + skip, P1 + skip);
                }, 2.0 * TGP_PW * m * (m + 1.0));
                if (rc) return rc;
            }
            if (early_next) {       // panel k+2's half of the next U2a (tile columns of blocks k+4, k+5), once that panel is complete
                TGP_HIP(tgp_await(ctx, st, 2, ctx->ev[6]));
                launch_syrk<1>(st, d_A, Np, k + 4, T3, 4, panel(k + 2) + (int64_t)2 * TGP_PW * TGP_PW, nullptr);
            }
            early = early_next;
            TGP_HIP(tgp_await(ctx, st, 1, ctx->ev[5]));
        }
        return 0;
    };
    if (mode == 0) {
        for (int k = 0; k < nP; ++k) {
            const int64_t mk = Np - (int64_t)TGP_PW * k;
            factor_panel(st, panel(k), mk, Wk(k), ctx->d_info, k * TGP_PW, false, n_data, psync(k), 256);
            const int T = (int)((mk - TGP_PW) / TGP_TB);
            if (T > 0) {
                const double m = (double)T * TGP_TB;
                int rc = timed([&] { launch_syrk<1>(st, d_A, Np, k + 1, T, 0, panel(k) + (int64_t)TGP_PW * TGP_PW, nullptr); },
                               (double)TGP_PW * m * (m + 1.0));
                if (rc) return rc;
            }
        }
    } else if (mode == 3) {
        // Groups of four panels: inside a group, panel j is brought up to date by one strip launch of depth 256 j
        // (its two tile columns against the j panels before it) and factored; everything right of the group is then
        // updated in one pass of depth 1024 -- U4a (the 8 tile columns of the next group) and U4b (the rest), the next
        // group running on the side stream under U4b.
        hipStream_t sd = ctx->side_stream;
        auto seg_rows = [&](int kpanel, int first_block) {       // row of panel `kpanel` that belongs to `first_block`
            return (const double *)(panel(kpanel) + (int64_t)(first_block - kpanel) * TGP_PW * TGP_PW);
        };
        auto strip_update = [&](hipStream_t s, int k0, int j) {  // panel k0+j's columns -= sum_{i<j} P_{k0+i}
            const int ob = k0 + j;
            const int T = (int)((Np - (int64_t)TGP_PW * ob) / TGP_TB);
            if (T <= small_t()) {
                for (int i = 0; i < j; ++i)
                    syrk_small_kernel<1><<<dim3((unsigned)(T * 8), 2u), 256, 0, s>>>(d_A, Np, ob, T, seg_rows(k0 + i, ob), nullptr);
                return;
            }
            const unsigned gs = (unsigned)((int64_t)T * 2);
            if (j == 1) {
                SegPtrs<1> P{{seg_rows(k0, ob)}, {nullptr}};
                syrk_segs_kernel<1><<<gs, 256, 0, s>>>(d_A, Np, ob, T, 2, P);
            } else if (j == 2) {
                SegPtrs<2> P{{seg_rows(k0, ob), seg_rows(k0 + 1, ob)}, {nullptr, nullptr}};
                syrk_segs_kernel<2><<<gs, 256, 0, s>>>(d_A, Np, ob, T, 2, P);
            } else {
                SegPtrs<3> P{{seg_rows(k0, ob), seg_rows(k0 + 1, ob), seg_rows(k0 + 2, ob)}, {nullptr, nullptr, nullptr}};
                syrk_segs_kernel<3><<<gs, 256, 0, s>>>(d_A, Np, ob, T, 2, P);
            }
        };
        auto factor_group = [&](hipStream_t s, int k0) {
            for (int j = 0; j < 4 && k0 + j < nP; ++j) {
                if (j > 0) strip_update(s, k0, j);
                factor_panel(s, panel(k0 + j), Np - (int64_t)TGP_PW * (k0 + j), Wk(k0 + j), ctx->d_info, (k0 + j) * TGP_PW, false, n_data);
            }
        };
        auto bulk = [&](int k0, int ob, int T, int strip) {      // depth-1024 update from block `ob` on, T tile rows
            if (T <= small_t()) {
                const int cols = strip == 0 ? T : (strip < T ? strip : T);
                for (int i = 0; i < 4; i += 2)
                    syrk_small_kernel<2><<<dim3((unsigned)(T * 8), (unsigned)cols), 256, 0, st>>>(
                        d_A, Np, ob, T, seg_rows(k0 + i, ob), seg_rows(k0 + i + 1, ob));
                return;
            }
            SegPtrs<4> P{{seg_rows(k0, ob), seg_rows(k0 + 1, ob), seg_rows(k0 + 2, ob), seg_rows(k0 + 3, ob)},
                         {nullptr, nullptr, nullptr, nullptr}};
            const unsigned gs = strip == 0 ? (unsigned)tilemap_grid(T) : (unsigned)((int64_t)T * strip);
            syrk_segs_kernel<4><<<gs, 256, 0, st>>>(d_A, Np, ob, T, strip, P);
        };
        // Below `tail_tiles` rows of trailing matrix the deeper grouping no longer pays (its strips and the longer
        // serial chain cost more than the per-tile overhead it saves: crossover at N ~ 24k): the tail runs in pairs.
        // (96 against 128 tile rows: 77.7 / 174.3 / 1321.8 ms against 78.1 / 174.7 / 1323.5 at N = 24 576 / 32 768 / 65 536; 64: 78.3 / 175.1)
        static const int tail_tiles = [] { const char *e = getenv("TGP_QUAD_TAIL_TILES"); return e ? atoi(e) : 96; }();
        factor_group(st, 0);
        for (int k = 0; k + 4 < nP; k += 4) {
            const int T4 = (int)((Np - (int64_t)TGP_PW * (k + 4)) / TGP_TB);        // tiles from block k+4
            if (T4 <= tail_tiles) {
                // hand-over: apply this group to everything right of it in one go, then continue pair-wise from k+4
                const double m = (double)T4 * TGP_TB;
                int rc = timed([&] { bulk(k, k + 4, T4, 0); }, 4.0 * TGP_PW * m * (m + 1.0));
                if (rc) return rc;
                counting = false;
                rc = run_pairs(k + 4);
                counting = true;
                if (rc) return rc;
                break;
            }
            {   // U4a: the 8 tile columns of the next group
                const double rows = (double)T4 * TGP_TB, w = (T4 < 8 ? T4 : 8) * (double)TGP_TB;
                const double elems = w * (rows - w) + w * (w + 1.0) / 2.0;
                int rc = timed([&] { bulk(k, k + 4, T4, 8); }, 2.0 * 4.0 * TGP_PW * elems);
                if (rc) return rc;
            }
            TGP_HIP(tgp_signal(ctx, st, 0, ctx->ev[4]));
            TGP_HIP(tgp_await(ctx, sd, 0, ctx->ev[4]));
            factor_group(sd, k + 4);
            TGP_HIP(tgp_signal(ctx, sd, 1, ctx->ev[5]));
            const int T5 = T4 - 8;
            if (T5 > 0) {   // U4b: everything from block k+8 on
                const double m = (double)T5 * TGP_TB;
                int rc = timed([&] { bulk(k, k + 8, T5, 0); }, 4.0 * TGP_PW * m * (m + 1.0));
                if (rc) return rc;
            }
            TGP_HIP(tgp_await(ctx, st, 1, ctx->ev[5]));
        }
    } else if (mode == 2) {
        int rc = run_pairs(0);
        if (rc) return rc;
    } else {
        for (int k = 0; k < nP; k += 2) {
            const int64_t mk = Np - (int64_t)TGP_PW * k;
            factor_panel(st, panel(k), mk, Wk(k), ctx->d_info, k * TGP_PW, false, n_data, psync(k), 256);
            if (k + 1 >= nP) break;
            const int T1 = (int)((mk - TGP_PW) / TGP_TB);
            // U1: only the two tile columns of panel k+1, depth 256 (short; not part of the timed set)
            launch_syrk<1>(st, d_A, Np, k + 1, T1, 2, panel(k) + (int64_t)TGP_PW * TGP_PW, nullptr);
            factor_panel(st, panel(k + 1), mk - TGP_PW, Wk(k + 1), ctx->d_info, (k + 1) * TGP_PW, false, n_data, psync(k + 1), 256);
            const int T2 = T1 - 2;
            if (T2 > 0) {
                const double m = (double)T2 * TGP_TB;
                int rc = timed([&] { launch_syrk<2>(st, d_A, Np, k + 2, T2, 0, panel(k) + (int64_t)2 * TGP_PW * TGP_PW,
                                                     panel(k + 1) + (int64_t)TGP_PW * TGP_PW); },
                               2.0 * TGP_PW * m * (m + 1.0));
                if (rc) return rc;
            }
        }
    }
    TGP_HIP(hipGetLastError());
    TGP_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    ctx->timings[6] = nlaunch;
    ctx->timings[7] = flops;
    ctx->timings[5] = 0.0;
    // `defer_info`: the caller queues more work behind the factorisation and reads *ctx->h_info after its own
    // synchronisation (one host round trip less per solve; the per-launch profile needs the synchronisation here)
    if (defer_info && !prof) return 0;       // (the caller maps a negative *h_info through tgp_potrf_info_rc)
    TGP_HIP(hipStreamSynchronize(st));
    if (prof) {
        double tot = 0.0;
        for (int i = 0; i < nlaunch; ++i) {
            float ms = 0.f;
            TGP_HIP(hipEventElapsedTime(&ms, ctx->prof_events[2 * i], ctx->prof_events[2 * i + 1]));
            tot += ms;
        }
        ctx->timings[5] = tot;
    }
    return tgp_potrf_info_rc(ctx, *ctx->h_info);
}

// what the device left in `info` -> return code: > 0 first failing pivot (1-based), 0 fine, -7 (an in-kernel hand-off of
// panel_mid_kernel gave up after its bounded wait) -> -4 with the reason in tgp_last_error
int tgp_potrf_info_rc(tgp_ctx *ctx, int info) {
    if (info >= 0) return info;
    ctx->err = "Cholesky: an in-kernel hand-off timed out (panel_mid_kernel waited 50 ms for workgroups of its own launch); "
               "the factor is incomplete -- TGP_PANEL_MID=0 selects the schedule without in-kernel waits";
    return TGP_RC_HANDOFF;
}

// 256x256 diagonal block (ld 256): L in place, inverses of its two 128-blocks to W0 / W1
// `latency`: the step between the two 128-blocks as two launches of eight 16-row slices (2 x ~8 us + a launch gap) instead of one
// workgroup's 42 us -- for the chain-bound phase; beside a long bulk launch every launch waits for slots and one is better
int launch_factor_diag256(tgp_ctx *ctx, double *blk, double *W0, double *W1, int base, bool latency) {
    hipStream_t st = ctx->stream;
    double *R1 = blk + (int64_t)TGP_TB * TGP_PW;
    const bool excl = ctx->chain_exclusive != 0;      // a queued bulk update keeps compute units clear for this chain
    run_potrf128(st, blk, TGP_PW, W0, ctx->d_info, base, excl);
    if (latency) {
        gemm_col_small_kernel<0, TGP_TB><<<8, 256, 0, st>>>(R1, W0, R1);
        gemm_col_small_kernel<1, TGP_PW><<<8, 256, 0, st>>>(R1, R1, R1 + TGP_TB);
    } else {
        diag_mid_kernel<<<1, 256, 0, st>>>(R1, W0);
    }
    run_potrf128(st, R1 + TGP_TB, TGP_PW, W1, ctx->d_info, base + TGP_TB, excl);
    TGP_HIP(hipGetLastError());
    return 0;
}

// rows (ntiles x 128, ld 256) <- rows L_kk^-T with L_kk given by its 256x256 block and W0, W1
// keepW != nullptr: [W0 | W1] (contiguous at W0) is also copied there
int launch_trsm_rows(tgp_ctx *ctx, double *rows, int ntiles, const double *Lkk, const double *W0, const double *W1, double *keepW) {
    if (ntiles <= 0) {
        if (keepW) TGP_HIP(hipMemcpyAsync(keepW, W0, (size_t)2 * TGP_TB * TGP_TB * 8, hipMemcpyDeviceToDevice, ctx->stream));
        return 0;
    }
    // up to 24 row tiles -- about where a rank's bulk update falls under the chain's time at 2 - 8 ranks: beside a long bulk
    // launch eight times the workgroups wait longer for slots than they save (sweep 0 / 16 / 24 / 32 / 48 / 64:
    // profiles/r04_chain_latency_ab.txt)
    constexpr int small_max = 24;
    if (ntiles <= small_max)
        panel_tall_small_kernel<<<ntiles * 8, 256, 0, ctx->stream>>>(rows, W0, Lkk + (int64_t)TGP_TB * TGP_PW, W1, keepW);
    else
        panel_tall_kernel<<<ntiles, 256, 0, ctx->stream>>>(rows, W0, Lkk + (int64_t)TGP_TB * TGP_PW, W1, keepW);
    TGP_HIP(hipGetLastError());
    return 0;
}

template <int NSEG>
static void launch_distn(hipStream_t st, unsigned grid, double *d_Aloc, const int64_t *d_loff, int kpanel, int G, int g,
                         const double *const *P, const int *cmax, int col_lo, int ncol, int nrows, const DistMap &M, int nres,
                         unsigned *queue, const double *const *PB) {
    DistSegs<NSEG> S;
    for (int s = 0; s < NSEG; ++s) {
        S.P[s] = P[s];
        S.PB[s] = PB ? PB[s] : P[s];
        S.cmax[s] = cmax[s];
        S.frg[s] = (int)dist_first_round(kpanel + s + 1, g, G);
    }
    if (nres > 0 && queue)
        syrk_distn_queue_kernel<NSEG><<<512 + 8, 256, 0, st>>>(d_Aloc, d_loff, kpanel, G, g, S, col_lo, ncol, nrows, M, grid / 8, nres, queue);
    else
        syrk_distn_kernel<NSEG><<<grid, 256, 0, st>>>(d_Aloc, d_loff, kpanel, G, g, S, col_lo, ncol, nrows, M);
}

// `head_cols` > 0: the fused form (DistMap) -- tile columns [col_lo, col_lo + head_cols) first, then the rest up to col_hi,
// one launch; tgp_head_flag() tells which flag / sequence number the launch publishes when the head is done.
// d_PB (optional): separate bases for the column operand of each segment (tgp_dd_strip_left: rows from this rank's own
// storage, columns from the broadcast)
int launch_syrk_distn(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                      const double *const *d_P, const int *cmax, int col_lo, int col_hi, int queue_nres, int head_cols,
                      const double *const *d_PB) {
    TGP_ARG(nseg >= 1 && nseg <= 4 && head_cols >= 0);
    const int64_t nB = Np / TGP_PW;
    const int64_t nloc = dist_panel_blocks(kpanel + nseg, nB, g, G);      // local blocks > kpanel + nseg - 1
    const int64_t ncol_all = 2 * (nB - kpanel - nseg);
    if (col_hi < 0 || col_hi > ncol_all) col_hi = (int)ncol_all;
    if (col_lo < 0) col_lo = 0;
    const int64_t ncol = (int64_t)col_hi - col_lo;
    hipStream_t st = ctx->stream;
    DistMap M;
    M.nhead = M.head_cols = M.whole = M.head_half = 0;
    M.half_from = 1 << 30;
    // what the panel chain waits for -- its own strips, the head of a fused launch -- runs as 64-row half tiles while the
    // halves still fit one round of slots (256 tiles): half the time per tile on the chain's critical path
    constexpr int half_max = 256;
    if (head_cols > 0) {
        // the waiter is released whatever this rank's share is (even none at all): the signal is part of the call's contract
        hipError_t e = hipSuccess;
        M.hs.seq = tgp_next_seq(ctx, TGP_FLAG_HEAD, &e);
        TGP_HIP(e);
        M.hs.flag = ctx->d_flags + 16 * TGP_FLAG_HEAD;
        M.hs.done = ctx->d_flags + 16 * TGP_FLAG_HEAD_COUNT;
    }
    if (nloc <= 0 || ncol <= 0) {
        if (head_cols > 0) TGP_HIP(tgp_signal_value(ctx, st, TGP_FLAG_HEAD, M.hs.seq));
        return 0;
    }
    // staircase of valid tiles: local tile row lt reaches global tile column gti(lt); super-tiles per group of 8 rows
    const int nrows = (int)(2 * nloc);
    const int64_t s0 = kpanel + nseg, qfb = dist_first_round(s0, g, G);
    M.qfb = (int)qfb;
    M.ginv = G == 1 ? 0u : (unsigned)((((uint64_t)1) << 32) / (uint64_t)G) + 1u;
    TGP_ARG(nB < 65536 && G >= 1);
    // Strips (the panel chain's updates of a few tile columns): a dense grid, one workgroup per (tile row, tile column).  The
    // super-tile map would start three empty workgroups for every useful one of a two-column strip, and beside a bulk
    // update that fills every slot it is slot grants, not arithmetic, that a strip waits for (rocprofv3, world of one at
    // N = 65 536: 748 / 561 / 375 us per strip of depth 768 / 512 / 256 against 468 / 345 / 239 us on the single-GPU path).
    unsigned grid = 0;
    if (head_cols == 0 && ncol <= 8 && queue_nres == 0) {
        M.ngroups = 0;
        M.start[0] = 0;
        M.head_half = (int64_t)nrows * ncol <= half_max ? 1 : 0;
        grid = (unsigned)((int64_t)nrows * ncol) << M.head_half;
    } else {
        if (head_cols > ncol) head_cols = (int)ncol;
        M.head_cols = head_cols;
        M.head_half = (head_cols > 0 && (int64_t)nrows * head_cols <= half_max) ? 1 : 0;
        M.nhead = ((nrows * head_cols << M.head_half) + 7) / 8 * 8;
        M.ngroups = (nrows + 7) / 8;
        TGP_ARG(M.ngroups <= 257);
        int total = 0;
        for (int R = 0; R < M.ngroups; ++R) {
            M.start[R] = total;
            const int ltmax = (R * 8 + 7 < nrows ? R * 8 + 7 : nrows - 1);
            const int64_t gti = 2 * (dist_block_of(qfb + (ltmax >> 1), g, G) - s0) + (ltmax & 1);     // last valid global column
            int64_t reach = gti - (col_lo + head_cols) + 1;                                  // valid columns of this launch
            if (reach > ncol - head_cols) reach = ncol - head_cols;
            total += reach > 0 ? (int)((reach + 7) / 8) : 0;
        }
        M.start[M.ngroups] = total;
        M.whole = 64 * (total / 8);
        const int slots = M.whole + 8 * (total % 8);          // tile slots per XCD class
        static const int half_tiles = [] { const char *e = getenv("TGP_DIST_HALF_TILES"); return e ? atoi(e) : 32; }();
        const int nhalf = slots < half_tiles ? slots : half_tiles;      // 32 per class = 256 tiles -> 512 half tiles = one round
        M.half_from = slots - nhalf;
        grid = (unsigned)(M.nhead + 8 * (slots + nhalf));
        if (grid == 0) return 0;
        if (head_cols > 0) {
            // every head workgroup counts itself in; the counter runs on from launch to launch (modulo 2^32: all of the
            // previous launch's increments are in before this one starts, same stream)
            ctx->head_count += (unsigned)M.nhead;
            M.hs.target = ctx->head_count;
        }
    }
    // queued form: one set of counters per launch, zeroed by tgp_dd_queue_reset at the start of a factorisation; beyond
    // TGP_NQUEUE launches (N > 131 072 in groups of four) the sets are reused in turn, each zeroed on this stream in front of
    // its launch -- the launch that used it TGP_NQUEUE launches ago is long finished (same stream)
    int nres = queue_nres > TGP_QUEUE_MAXRES ? TGP_QUEUE_MAXRES : queue_nres;
    unsigned *queue = nullptr;
    if (nres > 0 && M.ngroups != 0) {
        queue = ctx->d_queue + TGP_QUEUE_WORDS * (ctx->dist_nqueue % TGP_NQUEUE);
        if (ctx->dist_nqueue >= TGP_NQUEUE) TGP_HIP(hipMemsetAsync(queue, 0, TGP_QUEUE_WORDS * sizeof(unsigned), st));
        ++ctx->dist_nqueue;
    }
    switch (nseg) {
        case 1: launch_distn<1>(st, grid, d_Aloc, d_loff, kpanel, G, g, d_P, cmax, col_lo, (int)ncol, nrows, M, nres, queue, d_PB); break;
        case 2: launch_distn<2>(st, grid, d_Aloc, d_loff, kpanel, G, g, d_P, cmax, col_lo, (int)ncol, nrows, M, nres, queue, d_PB); break;
        case 3: launch_distn<3>(st, grid, d_Aloc, d_loff, kpanel, G, g, d_P, cmax, col_lo, (int)ncol, nrows, M, nres, queue, d_PB); break;
        default: launch_distn<4>(st, grid, d_Aloc, d_loff, kpanel, G, g, d_P, cmax, col_lo, (int)ncol, nrows, M, nres, queue, d_PB); break;
    }
    TGP_HIP(hipGetLastError());
    return 0;
}

// measurement hook: the depth-512 trailing update of a whole Np x Np packed matrix (as after the first
// panel pair), `reps` launches back to back on whatever d_A holds; average launch time by HIP events
int tgp_debug_syrk_loop(tgp_ctx *ctx, double *d_A, int64_t Np, int reps, double *ms_per_launch, double *flops_per_launch) {
    TGP_ARG(d_A && Np >= 4 * TGP_PW && Np % TGP_PW == 0 && reps > 0 && ms_per_launch && flops_per_launch);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int T2 = (int)((Np - (int64_t)TGP_PW * 2) / TGP_TB);
    const double *P0 = d_A + panel_off(0, Np) + (int64_t)2 * TGP_PW * TGP_PW;
    const double *P1 = d_A + panel_off(1, Np) + (int64_t)TGP_PW * TGP_PW;
    if (getenv("TGP_DEBUG_SEGS")) {           // the depth-1024 kernel on the same two panels twice
        SegPtrs<4> P{{P0, P1, P0, P1}, {nullptr, nullptr, nullptr, nullptr}};
        const unsigned gs = (unsigned)tilemap_grid(T2);
        syrk_segs_kernel<4><<<gs, 256, 0, st>>>(d_A, Np, 2, T2, 0, P);
        TGP_HIP(hipEventRecord(ctx->ev[0], st));
        for (int r = 0; r < reps; ++r) syrk_segs_kernel<4><<<gs, 256, 0, st>>>(d_A, Np, 2, T2, 0, P);
        TGP_HIP(hipEventRecord(ctx->ev[1], st));
        TGP_HIP(hipStreamSynchronize(st));
        float ms2 = 0.f;
        TGP_HIP(hipEventElapsedTime(&ms2, ctx->ev[0], ctx->ev[1]));
        const double m2 = (double)T2 * TGP_TB;
        *ms_per_launch = ms2 / reps;
        *flops_per_launch = 4.0 * TGP_PW * m2 * (m2 + 1.0);
        return 0;
    }
    launch_syrk<2>(st, d_A, Np, 2, T2, 0, P0, P1);
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    for (int r = 0; r < reps; ++r) launch_syrk<2>(st, d_A, Np, 2, T2, 0, P0, P1);
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    const double m = (double)T2 * TGP_TB;
    *ms_per_launch = ms / reps;
    *flops_per_launch = 2.0 * TGP_PW * m * (m + 1.0);
    return 0;
}

#ifdef TGP_POTRF_STAMPS
extern "C" int tgp_debug_gemm_stamps(unsigned long long *out, int grid) {
    if (grid > 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(tgp_gemm_stamp_grid), &grid, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_gemm_stamps), 1024 * 4 * sizeof(unsigned long long));
}
extern "C" int tgp_debug_queue_stamps(unsigned long long *out, int T) {
    if (T > 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(tgp_queue_stamp_T), &T, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_queue_stamps), 1024 * 4 * sizeof(unsigned long long));
}
extern "C" int tgp_debug_mid_stamps(unsigned long long *out, int base) {
    if (base >= 0) return (int)hipMemcpyToSymbol(HIP_SYMBOL(tgp_mid_stamp_base), &base, sizeof(int));
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_mid_stamps), 25 * 8 * sizeof(unsigned long long));
}
extern "C" int tgp_debug_potrf_fine(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_potrf_fine), 1024 * 8 * sizeof(unsigned long long));      // [block][sub-step]
}
extern "C" int tgp_debug_potrf_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(tgp_potrf_stamps), 1024 * 20 * sizeof(unsigned long long));   // [block][stamp]
}
#endif

int tgp_debug_tilemap(int64_t T, int32_t *ti, int32_t *tj, int64_t cap) {
    const int64_t g = tilemap_grid(T);
    if (cap < g) return -1;
    for (int64_t b = 0; b < g; ++b) {
        int i, j;
        tilemap(b, T, i, j);
        ti[b] = i;
        tj[b] = j;
    }
    return (int)g;
}
