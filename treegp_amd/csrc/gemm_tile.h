// The NT fp64 MFMA tile shared by the Cholesky (chol.hip) and the multi-GPU kernels (dist.hip):
// C (128x128) = or -= A (128 x KDEPTH) * B (128 x KDEPTH)^T, all row-major.
//
// 128x128 per workgroup of 4 waves (2x2, 64x64 per wave = 4x4 v_mfma_f64_16x16x4_f64 tiles, 128
// accumulator VGPRs); operands staged through LDS in 16-deep k-chunks with a row stride of 18
// doubles (conflict-free ds_read_b64 fragment reads: lane (r, kq) -> slot 18 r + kq mod 32 is a
// permutation), double-buffered, the next chunk's global loads in flight during the MFMAs.
// A and C always live in 256-wide panels (ld 256); B is a panel (LDB 256) or a W block (LDB 128).
#pragma once
#include "tgp_internal.h"

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int KB = 16;         // k-chunk depth

// Variant knobs (A/B-measured on MI355X, see LAB_NOTES.md A.4):
//   LSV   LDS row stride in doubles.  hipcc fuses the fragment reads of two k-steps into
//         ds_read2_b64, whose banking is per 16-lane group with (addr/4) mod 32: an odd stride
//         (17) is conflict-free there, the even stride 18 is 2-way.  Odd stride means 8-byte
//         aligned rows, so staging stores are ds_write_b64 pairs.
//   PRE   MODE 1 only: load C into the accumulators (negated) in the prologue, together with the
//         first operand chunk (one memory latency), instead of a 4-round read-modify-write epilogue.
//   (An LDS-DMA variant of the staging -- `buffer_load ... lds`, XOR-swizzled source side -- was measured slower, 59.7 vs
//   63.0 TF on the depth-512 update, and removed in round 4: LAB_NOTES.md A.4, git history.)
template <int LSV, bool PRE>
struct TileCfg {
    static constexpr int LS = LSV;
    static constexpr bool PRELOAD = PRE;
};
using TileDefault = TileCfg<17, true>;

// buffer addressing (one wave-uniform 128-bit descriptor per operand, one 32-bit lane offset, the
// row/column part of every access as a scalar offset): keeps the 64 C accesses and the staging
// loads from each needing their own 64-bit address VGPR pair.
using v2u = __attribute__((ext_vector_type(2))) unsigned;
using v4u = __attribute__((ext_vector_type(4))) unsigned;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tile_rsrc(const double *p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double buf_ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ double2 buf_ld2(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(double2, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st1(double x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, soff, 0);
}
// the same with the non-temporal bit (streaming: the line is the first to be evicted) -- for the C tile of the trailing update,
// touched once per launch, so that it does not push operand panels out of the XCD's L2 (depth-1024 update: Cholesky
// +0.4 ... 0.8 % at N = 22 528 ... 65 536, A/B builds on one box; no effect on the depth-512 tile, which keeps plain accesses)
#ifndef TGP_C_NT
#define TGP_C_NT 1
#endif
__device__ __forceinline__ double buf_ld1_stream(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, TGP_C_NT ? 2 : 0));
}
__device__ __forceinline__ void buf_st1_stream(double x, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x), r, voff, soff, TGP_C_NT ? 2 : 0);
}

template <int LS>
__device__ __forceinline__ void lds_put2(double *p, const double2 &v) {
    if constexpr ((LS & 1) == 0) {
        *reinterpret_cast<double2 *>(p) = v;
    } else {
        p[0] = v.x;
        p[1] = v.y;
    }
}

// MODE 0: C = A B^T     MODE 1: C -= A B^T     MODE 2: C = -(A B^T)      (A: 128 x KDEPTH, B: 128 x KDEPTH, row-major)
// A and C always live in 256-wide panels (ld 256); B is a panel (LDB 256) or a W block (LDB 128).
// NSEG = 2: the contraction runs over two operand pairs back to back, C -= A0 B0^T + A1 B1^T (the
// depth-512 trailing update after two factored panels), one chunk pipeline across both.
// NSEG = 0: run-time number of segments `nseg_rt`, segment s at a_ptr + s * seg_stride_a (and likewise
// for B) -- the posterior-covariance product over all panels of the factor.
template <int LS>
__device__ __forceinline__ double *tile128_lds_storage() {
    __shared__ __attribute__((aligned(16))) double lds[2 * 2 * 128 * LS];
    return lds;
}

// gemm_tile_128_at: the tile with its staging buffers at `lds_base` (2 * 2 * 128 * LS doubles) -- for kernels that alias them
// over another LDS image (chol.hip: panel_mid_kernel, whose diagonal-block workgroup holds potrf128's 96 KB there);
// gemm_tile_128: the same on the kernel's own static array.
template <int MODE, int LDB, int KDEPTH, typename CFG = TileDefault, int NSEG = 1>
__device__ __forceinline__ void gemm_tile_128_at(double *lds_base, const double *a_ptr, const double *b_ptr, double *c_ptr,
                                                 const double *a1_ptr = nullptr, const double *b1_ptr = nullptr, int nseg_rt = 1,
                                                 int64_t seg_stride_a = 0, int64_t seg_stride_b = 0) {
    constexpr int LDA = TGP_PW, LDC = TGP_PW;
    constexpr int LS = CFG::LS;
    constexpr bool PRELOAD = CFG::PRELOAD && MODE == 1;
    // [buf][A|B][row*LS + k]; one array per LS, shared by every MODE / LDB instantiation a kernel calls in sequence
    double (*lds)[2][128 * LS] = reinterpret_cast<double (*)[2][128 * LS]>(lds_base);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int wr = w >> 1, wc = w & 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    // staging map: piece s of this thread = row (tid>>3) + 32 s, doubles kp..kp+1
    const int srow = tid >> 3;
    const int kp = (tid & 7) * 2;
    const __amdgpu_buffer_rsrc_t ra_src = tile_rsrc(a_ptr, 128 * LDA * 8);
    const __amdgpu_buffer_rsrc_t rb_src = tile_rsrc(b_ptr, 128 * LDB * 8);
    const __amdgpu_buffer_rsrc_t ra1_src = tile_rsrc(NSEG > 1 ? a1_ptr : a_ptr, 128 * LDA * 8);
    const __amdgpu_buffer_rsrc_t rb1_src = tile_rsrc(NSEG > 1 ? b1_ptr : b_ptr, 128 * LDB * 8);
    const __amdgpu_buffer_rsrc_t rc_dst = tile_rsrc(c_ptr, 128 * LDC * 8);
    const int va = (srow * LDA + kp) * 8, vb = (srow * LDB + kp) * 8;       // per-lane byte offsets

    double2 ra[4], rb[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        ra[s] = buf_ld2(ra_src, va, s * 32 * LDA * 8);
        rb[s] = buf_ld2(rb_src, vb, s * 32 * LDB * 8);
    }

    // C fragment map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 r
    const int vc = ((wr * 64 + l4) * LDC + wc * 64 + l15) * 8;
    d4 acc[4][4];
    if constexpr (PRELOAD) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][n][r] = buf_ld1(rc_dst, vc, ((m * 16 + 4 * r) * LDC + n * 16) * 8);
    } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0.0, 0.0, 0.0, 0.0};
    }

#pragma unroll
    for (int s = 0; s < 4; ++s) {
        lds_put2<LS>(&lds[0][0][(srow + 32 * s) * LS + kp], ra[s]);
        lds_put2<LS>(&lds[0][1][(srow + 32 * s) * LS + kp], rb[s]);
    }
    __syncthreads();
    if constexpr (PRELOAD) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = -acc[m][n];
    }

    constexpr int cps = KDEPTH / KB;               // chunks per segment
    const int nchunk = (NSEG > 0 ? NSEG : nseg_rt) * cps;
    const int fa = (wr * 64 + l15) * LS + l4;      // fragment read offsets
    const int fb = (wc * 64 + l15) * LS + l4;
    for (int c = 0; c < nchunk; ++c) {
        const int buf = c & 1;
        const bool more = (c + 1 < nchunk);
        if (more) {
            const int cn = c + 1;
            const int k0 = (NSEG != 1 ? (cn % cps) : cn) * KB;
            const bool second = NSEG > 1 && cn >= cps;           // wave-uniform
            __amdgpu_buffer_rsrc_t sa = second ? ra1_src : ra_src;
            __amdgpu_buffer_rsrc_t sb = second ? rb1_src : rb_src;
            if constexpr (NSEG == 0) {
                const int seg = cn / cps;
                sa = tile_rsrc(a_ptr + seg * seg_stride_a, 128 * LDA * 8);
                sb = tile_rsrc(b_ptr + seg * seg_stride_b, 128 * LDB * 8);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ra[s] = buf_ld2(sa, va, (s * 32 * LDA + k0) * 8);
                rb[s] = buf_ld2(sb, vb, (s * 32 * LDB + k0) * 8);
            }
        }
        const double *As = lds[buf][0];
        const double *Bs = lds[buf][1];
#pragma unroll
        for (int k4 = 0; k4 < KB / 4; ++k4) {
            double af[4], bf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) af[m] = As[fa + m * 16 * LS + k4 * 4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = Bs[fb + n * 16 * LS + k4 * 4];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
        // keep the staging stores (which wait for the global loads) and the barrier BEHIND the
        // chunk's MFMAs: hipcc otherwise hoists them above two thirds of the MFMAs and every chunk
        // stalls on memory latency
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                lds_put2<LS>(&lds[buf ^ 1][0][(srow + 32 * s) * LS + kp], ra[s]);
                lds_put2<LS>(&lds[buf ^ 1][1][(srow + 32 * s) * LS + kp], rb[s]);
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int m = 0; m < 4; ++m) {
        if constexpr (MODE == 1 && !PRELOAD) {
            double old[4][4];
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) old[n][r] = buf_ld1(rc_dst, vc, ((m * 16 + 4 * r) * LDC + n * 16) * 8);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    buf_st1(old[n][r] - acc[m][n][r], rc_dst, vc, ((m * 16 + 4 * r) * LDC + n * 16) * 8);
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    buf_st1((PRELOAD || MODE == 2) ? -acc[m][n][r] : acc[m][n][r], rc_dst, vc, ((m * 16 + 4 * r) * LDC + n * 16) * 8);
        }
    }
}

template <int MODE, int LDB, int KDEPTH, typename CFG = TileDefault, int NSEG = 1>
__device__ __forceinline__ void gemm_tile_128(const double *a_ptr, const double *b_ptr, double *c_ptr,
                                              const double *a1_ptr = nullptr, const double *b1_ptr = nullptr, int nseg_rt = 1,
                                              int64_t seg_stride_a = 0, int64_t seg_stride_b = 0) {
    gemm_tile_128_at<MODE, LDB, KDEPTH, CFG, NSEG>(tile128_lds_storage<CFG::LS>(), a_ptr, b_ptr, c_ptr, a1_ptr, b1_ptr, nseg_rt,
                                                   seg_stride_a, seg_stride_b);
}


// ---- trailing-update tile, second generation ("DTV"): A straight into VGPRs, only B through LDS ----------
// C (32 NW x 128) -= sum over NSEG operand pairs of A (32 NW x KDEPTH) B (128 x KDEPTH)^T, everything in
// 256-wide panels (ld 256).  NW waves stacked along M: wave w owns rows 32w .. 32w+31 and all 128 columns
// (2 x 8 MFMA tiles, 128 accumulator VGPRs).  Its A rows are private, so A fragments are loaded from global
// memory directly in MFMA layout (double-buffered in registers) and never touch LDS; B (shared by all waves)
// is staged through LDS once per workgroup.  Measured motive: on the 2x2-wave tile above the VGPR->LDS
// staging stores alone cost 7 % (63.0 -> 67.7 TF with them removed); here they are 1/2 (NW = 4) or 1/4
// (NW = 8, 256 x 128 per workgroup) of that per flop.
// k assignment: lane (r, kq) handles k = 8h + 2kq + {0, 1} in MFMA steps 2h, 2h+1, so both the direct A
// loads and the B fragment reads are 16-byte accesses.  B rows are padded to 18 doubles: that leaves every ds_read_b128
// 2-way conflicted under gfx950's lane groups ({0-3, 12-15, 20-27}, ...: SQ_LDS_BANK_CONFLICT = 0.40 of SQ_LDS_IDX_ACTIVE,
// profiles/r04_pmc_sq.txt); a stride of 20 is conflict-free and was measured -- no difference (1293-1296 ms of trailing update
// per N = 65 536 solve either way, profiles/r04_lds_stride_ab.txt): the LDS array is 14 % busy and the MFMA pipe 92.6 %, the
// waves wait for the pipe, not for B.  18 stays: 4 KB less LDS per workgroup.
// NSEG = 0: run-time number of operand pairs, pair s at a_ptr + s * seg_stride_a / b_ptr + s * seg_stride_b.
// B staging buffers of the DTV tiles: ONE array per kernel whatever mix of tile variants it instantiates (a __shared__
// array inside a template is one per instantiation: two of them would push a trailing-update workgroup from 37 to 74 KB and
// potrf128's 96 KB would no longer fit on a compute unit beside it)
constexpr int DTV_LSB = 18;
__device__ __forceinline__ double *dtv_lds_storage() {
    __shared__ __attribute__((aligned(16))) double lds[2 * 128 * DTV_LSB];
    return lds;
}

template <int NW, int KDEPTH, int NSEG, int MT = 2>
__device__ __forceinline__ void gemm_tile_dtv(const double *a_ptr, const double *b_ptr, double *c_ptr,
                                              const double *a1_ptr, const double *b1_ptr, int nseg_rt = 1,
                                              int64_t seg_stride_a = 0, int64_t seg_stride_b = 0) {
    constexpr int LD = TGP_PW;
    constexpr int LSB = DTV_LSB;
    constexpr int BPT = 16 / NW;                    // B staging pieces (16 B) per thread and chunk
    constexpr int BROWS = 8 * NW;                   // rows covered by one staging pass
    double (*ldsB)[128 * LSB] = reinterpret_cast<double (*)[128 * LSB]>(dtv_lds_storage());

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    const __amdgpu_buffer_rsrc_t ra0 = tile_rsrc(a_ptr, 16 * MT * NW * LD * 8);
    const __amdgpu_buffer_rsrc_t rb0 = tile_rsrc(b_ptr, 128 * LD * 8);
    const __amdgpu_buffer_rsrc_t ra1 = tile_rsrc(NSEG > 1 ? a1_ptr : a_ptr, 16 * MT * NW * LD * 8);
    const __amdgpu_buffer_rsrc_t rb1 = tile_rsrc(NSEG > 1 ? b1_ptr : b_ptr, 128 * LD * 8);
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(c_ptr, 16 * MT * NW * LD * 8);
    const int va = ((16 * MT * w + l15) * LD + 2 * l4) * 8;          // A: row of m-tile 0, k-pair kq
    const int srow = tid >> 3, kp = (tid & 7) * 2;
    const int vb = (srow * LD + kp) * 8;                        // B staging piece
    const int vc = ((16 * MT * w + l4) * LD + l15) * 8;              // C fragment: col = lane & 15, row = (lane >> 4) + 4r
    const int fb = l15 * LSB + 2 * l4;                          // B fragment read

    double2 areg[2][MT][2];                                      // [set][m][h]
    double2 rbst[BPT];
    auto load_a = [&](double2 (&dst)[MT][2], __amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) dst[m][h] = buf_ld2(src, va, (m * 16 * LD + k0 + 8 * h) * 8);
    };
    auto load_b = [&](__amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) rbst[s] = buf_ld2(src, vb, (s * BROWS * LD + k0) * 8);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) *reinterpret_cast<double2 *>(&ldsB[buf][(srow + BROWS * s) * LSB + kp]) = rbst[s];
    };
    load_a(areg[0], ra0, 0);
    load_b(rb0, 0);

    d4 acc[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = buf_ld1(rc, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8);
    store_b(0);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = -acc[m][n];

    constexpr int cps = KDEPTH / KB;
    static_assert(cps % 2 == 0, "chunks are processed in register-set pairs");
    const int nchunk = (NSEG > 0 ? NSEG : nseg_rt) * cps;
    auto step = [&](const int c, double2 (&cur)[MT][2], double2 (&nxt)[MT][2]) {
        const int buf = c & 1;
        const bool more = (c + 1 < nchunk);
        if (more) {
            const int cn = c + 1;
            const int k0 = (cn % cps) * KB;
            const bool second = NSEG > 1 && cn >= cps;           // wave-uniform
            __amdgpu_buffer_rsrc_t sa = second ? ra1 : ra0, sb = second ? rb1 : rb0;
            if constexpr (NSEG == 0) {
                const int seg = cn / cps;
                sa = tile_rsrc(a_ptr + seg * seg_stride_a, 16 * MT * NW * LD * 8);
                sb = tile_rsrc(b_ptr + seg * seg_stride_b, 128 * LD * 8);
            }
            load_a(nxt, sa, k0);
            load_b(sb, k0);
        }
        const double *Bs = ldsB[buf];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 bf[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) bf[n] = *reinterpret_cast<const double2 *>(&Bs[fb + n * 16 * LSB + 8 * h]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].x, bf[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].y, bf[n].y, acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);          // staging stores (they wait for the loads) behind the MFMAs
        if (more) store_b(buf ^ 1);
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
            for (int r = 0; r < 4; ++r) buf_st1(-acc[m][n][r], rc, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8);
}


// ---- the DTV tile for the chain's strips: 32 x 128 of C per workgroup -------------------------------------------------------
// Waves 2 (rows) x 2 (column halves): wave (wr, wc) owns rows 16 wr .. 16 wr + 15 and columns 64 wc .. 64 wc + 63 (1 x 4 MFMA
// tiles).  Same staging of B, same k order per element as gemm_tile_dtv -- the results have the same bits -- but a quarter of
// the 128 x 128 tile's matrix instructions per wave: a strip that the panel chain waits for is a single round of workgroups,
// and its duration is one workgroup's (depth 256: 64 x 128 tiles ~22 us, these ~15).  Twice the B traffic per flop of the
// 64-row tile; the strips' operands sit in L2.
template <int KDEPTH, int NSEG>
__device__ __forceinline__ void gemm_tile_dtv32(const double *a_ptr, const double *b_ptr, double *c_ptr, const double *a1_ptr,
                                                const double *b1_ptr) {
    static_assert(NSEG == 1 || NSEG == 2, "one or two operand pairs");
    constexpr int LD = TGP_PW;
    constexpr int LSB = DTV_LSB;
    constexpr int BPT = 4, BROWS = 32;
    double (*ldsB)[128 * LSB] = reinterpret_cast<double (*)[128 * LSB]>(dtv_lds_storage());
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = w & 1, wc = w >> 1;
    const int l15 = lane & 15, l4 = lane >> 4;

    const __amdgpu_buffer_rsrc_t ra0 = tile_rsrc(a_ptr, 32 * LD * 8);
    const __amdgpu_buffer_rsrc_t rb0 = tile_rsrc(b_ptr, 128 * LD * 8);
    const __amdgpu_buffer_rsrc_t ra1 = tile_rsrc(NSEG > 1 ? a1_ptr : a_ptr, 32 * LD * 8);
    const __amdgpu_buffer_rsrc_t rb1 = tile_rsrc(NSEG > 1 ? b1_ptr : b_ptr, 128 * LD * 8);
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(c_ptr, 32 * LD * 8);
    const int va = ((16 * wr + l15) * LD + 2 * l4) * 8;
    const int srow = tid >> 3, kp = (tid & 7) * 2;
    const int vb = (srow * LD + kp) * 8;
    const int vc = ((16 * wr + l4) * LD + 64 * wc + l15) * 8;
    const int fb = (64 * wc + l15) * LSB + 2 * l4;

    double2 areg[2][2];                                          // [set][h]
    double2 rbst[BPT];
    auto load_a = [&](double2 (&dst)[2], __amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) dst[h] = buf_ld2(src, va, (k0 + 8 * h) * 8);
    };
    auto load_b = [&](__amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) rbst[s] = buf_ld2(src, vb, (s * BROWS * LD + k0) * 8);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) *reinterpret_cast<double2 *>(&ldsB[buf][(srow + BROWS * s) * LSB + kp]) = rbst[s];
    };
    load_a(areg[0], ra0, 0);
    load_b(rb0, 0);
    d4 acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = buf_ld1(rc, vc, (4 * r * LD + n * 16) * 8);
    store_b(0);
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = -acc[n];

    constexpr int cps = KDEPTH / KB;
    static_assert(cps % 2 == 0, "chunks are processed in register-set pairs");
    constexpr int nchunk = NSEG * cps;
    auto step = [&](const int c, double2 (&cur)[2], double2 (&nxt)[2]) {
        const int buf = c & 1;
        const bool more = (c + 1 < nchunk);
        if (more) {
            const int cn = c + 1;
            const int k0 = (cn % cps) * KB;
            const bool second = NSEG > 1 && cn >= cps;           // wave-uniform
            load_a(nxt, second ? ra1 : ra0, k0);
            load_b(second ? rb1 : rb0, k0);
        }
        const double *Bs = ldsB[buf];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 bf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bf[n] = *reinterpret_cast<const double2 *>(&Bs[fb + n * 16 * LSB + 8 * h]);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[h].x, bf[n].x, acc[n], 0, 0, 0);
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[h].y, bf[n].y, acc[n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_b(buf ^ 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int c = 0; c < nchunk; c += 2) {
        step(c, areg[0], areg[1]);
        step(c + 1, areg[1], areg[0]);
    }
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) buf_st1(-acc[n][r], rc, vc, (4 * r * LD + n * 16) * 8);
}


// ---- latency tile for small problems ----------------------------------------------------------------------
// One workgroup = 16 rows x 128 columns of C, wave w the columns 32w .. 32w+31 (1 x 2 MFMA tiles, two
// independent accumulator chains).  No LDS: A (16 x K) and B (128 x K) fragments come straight from global
// memory / L2 in MFMA layout with the k-permuted 16-byte loads of the DTV tile.  8x the L2 traffic per flop
// of the 128 x 128 tiles, but a 128 x 128 x K product becomes 32 waves' worth of parallel work instead of
// one workgroup: what a factorisation with only a few tiles per step needs (N below a few thousand).
// The barrier before the epilogue makes in-place use (C aliasing A) safe, as in the panel solve.
template <int MODE, int KDEPTH, int NSEG>
__device__ __forceinline__ void nt_small_tile(const double *a, int lda, const double *b, int ldb, double *c, int ldc,
                                              const double *a1, const double *b1) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    constexpr int CH = KDEPTH / KB;
    constexpr int UN = CH < 4 ? CH : 4;                 // chunks whose loads are issued together
#pragma unroll
    for (int seg = 0; seg < NSEG; ++seg) {
        const double *ap = (seg ? a1 : a) + (int64_t)l15 * lda + 2 * l4;
        const double *bp = (seg ? b1 : b) + (int64_t)(32 * w + l15) * ldb + 2 * l4;
#pragma unroll 1
        for (int c0 = 0; c0 < CH; c0 += UN) {
            double2 af[UN][2], bf[UN][2][2];
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int k = (c0 + u) * KB + 8 * h;
                    af[u][h] = *reinterpret_cast<const double2 *>(ap + k);
                    bf[u][0][h] = *reinterpret_cast<const double2 *>(bp + k);
                    bf[u][1][h] = *reinterpret_cast<const double2 *>(bp + (int64_t)16 * ldb + k);
                }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].x, bf[u][0][h].x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].x, bf[u][1][h].x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].y, bf[u][0][h].y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[u][h].y, bf[u][1][h].y, acc[1], 0, 0, 0);
                }
        }
    }
    __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *p = c + (int64_t)(l4 + 4 * r) * ldc + 32 * w + 16 * n + l15;
            if constexpr (MODE == 1) *p = *p - acc[n][r];
            else if constexpr (MODE == 2) *p = -acc[n][r];
            else *p = acc[n][r];
        }
}


// ---- the latency tile staged through LDS ------------------------------------------------------------------------------------
// The same 16 x 128 slice of C, the same matrix instructions in the same order (same bits), but the operands are fetched with
// whole-row loads -- a wave's load instruction covers two 512-byte row pieces instead of 16-byte pieces of 16 different rows --
// into registers, written to LDS in blocks of 64 k (B 128 x 64, A 16 x 64, row stride 66 doubles: the fragment reads of eight
// consecutive lanes fall into eight different 16-byte bank groups) and read back in MFMA layout.  The direct-from-global form
// above spends 6.7 us on a K = 128 slice whose matrix instructions take 1.7 (in-kernel stamps, profiles/r05_mid_stamps.txt),
// with all its loads in flight at once or not: the vector-memory pipeline's rate on 192 wave instructions that touch 16 cache
// lines each.  `lds`: NT_SLICE_LDS_DOUBLES doubles, 16-byte aligned, not used by anyone else during the call.
typedef double d2v __attribute__((ext_vector_type(2)));
constexpr int NT_SLICE_LS = 66;
constexpr int NT_SLICE_LDS_DOUBLES = (128 + 16) * NT_SLICE_LS;
// NCOL = 64: a 16 x 64 slice (wave w the columns 16 w .. 16 w + 15, one accumulator chain): half the matrix instructions and
// half of B per workgroup, for the steps in which twice the workgroups find a compute unit each (panel_mid_kernel).
template <int MODE, int KDEPTH, int NSEG, int NCOL = 128>
__device__ __forceinline__ void nt_slice_tile(double *lds, const double *a, int lda, const double *b, int ldb, double *c, int ldc,
                                              const double *a1, const double *b1) {
    constexpr int LS = NT_SLICE_LS, KBLK = 64;
    constexpr int BPS = KDEPTH / KBLK, NB = NSEG * BPS;          // blocks per segment / in all
    static_assert(KDEPTH % KBLK == 0 && NB >= 1, "whole blocks of 64 k");
    static_assert(NCOL == 128 || NCOL == 64, "slices of 128 or 64 columns");
    constexpr int NRB = NCOL / 8, NT = NCOL / 64, WC = NCOL / 4;      // B pieces per thread and block / MFMA tiles per wave / columns per wave
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int prow = t >> 5, pc = 2 * (t & 31);                  // this thread's 16-byte piece: rows prow + 8 i, doubles pc, pc + 1 of the block
    d4 acc[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
    d2v rb0[NRB], ra0[2], rb1[NRB], ra1[2];                    // two blocks in flight (named sets, macros: no dynamic register indexing)
#define NT_SLICE_FETCH(RB, RA, BLK)                                                                                    \
    {                                                                                                                  \
        constexpr int seg_ = (BLK) / BPS, k0_ = ((BLK) - seg_ * BPS) * KBLK;                                           \
        const double *bp_ = (seg_ ? b1 : b) + k0_ + pc, *ap_ = (seg_ ? a1 : a) + k0_ + pc;                             \
        _Pragma("unroll") for (int i = 0; i < NRB; ++i) RB[i] = *reinterpret_cast<const d2v *>(bp_ + (int64_t)(prow + 8 * i) * ldb); \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) RA[i] = *reinterpret_cast<const d2v *>(ap_ + (int64_t)(prow + 8 * i) * lda);  \
    }
#define NT_SLICE_STAGE(RB, RA)                                                                                         \
    {                                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NRB; ++i) *reinterpret_cast<d2v *>(lds + (prow + 8 * i) * LS + pc) = RB[i];             \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) *reinterpret_cast<d2v *>(lds + (128 + prow + 8 * i) * LS + pc) = RA[i];        \
    }
    auto multiply = [&] {
        const double *As = lds + (128 + l15) * LS + 2 * l4;
        const double *B0 = lds + (WC * w + l15) * LS + 2 * l4, *B1 = B0 + 16 * LS;
#pragma unroll
        for (int u = 0; u < KBLK / KB; ++u)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = u * KB + 8 * h;
                const d2v af = *reinterpret_cast<const d2v *>(As + k);
                const d2v b0 = *reinterpret_cast<const d2v *>(B0 + k);
                if constexpr (NT == 2) {
                    const d2v b1v = *reinterpret_cast<const d2v *>(B1 + k);
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.x, b0.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.x, b1v.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.y, b0.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.y, b1v.y, acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.x, b0.x, acc[0], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(af.y, b0.y, acc[0], 0, 0, 0);
                }
            }
    };
    // blocks 2 j (set 0) and 2 j + 1 (set 1), j = 0 .. 3: written out, the block numbers are compile-time constants
#define NT_SLICE_PAIR(J)                                                                                               \
    if constexpr (2 * (J) < NB) {                                                                                      \
        if constexpr ((J) > 0) __syncthreads();                                                                        \
        NT_SLICE_STAGE(rb0, ra0)                                                                                       \
        if constexpr (2 * (J) + 2 < NB) NT_SLICE_FETCH(rb0, ra0, 2 * (J) + 2)                                          \
        __syncthreads();                                                                                               \
        multiply();                                                                                                    \
        if constexpr (2 * (J) + 1 < NB) {                                                                              \
            __syncthreads();                                                                                           \
            NT_SLICE_STAGE(rb1, ra1)                                                                                   \
            if constexpr (2 * (J) + 3 < NB) NT_SLICE_FETCH(rb1, ra1, 2 * (J) + 3)                                      \
            __syncthreads();                                                                                           \
            multiply();                                                                                                \
        }                                                                                                              \
    }
    static_assert(NB <= 8, "K up to 256, one or two operand pairs");
    NT_SLICE_FETCH(rb0, ra0, 0)
    if constexpr (NB > 1) NT_SLICE_FETCH(rb1, ra1, 1)
    NT_SLICE_PAIR(0)
    NT_SLICE_PAIR(1)
    NT_SLICE_PAIR(2)
    NT_SLICE_PAIR(3)
#undef NT_SLICE_PAIR
#undef NT_SLICE_STAGE
#undef NT_SLICE_FETCH
    __syncthreads();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double *p = c + (int64_t)(l4 + 4 * r) * ldc + WC * w + 16 * n + l15;
            if constexpr (MODE == 1) *p = *p - acc[n][r];
            else if constexpr (MODE == 2) *p = -acc[n][r];
            else *p = acc[n][r];
        }
}
__device__ __forceinline__ double *nt_slice_lds_storage() {
    __shared__ __attribute__((aligned(16))) double lds[NT_SLICE_LDS_DOUBLES];
    return lds;
}

// The same tile for a compile-time list of NSEG operand pairs (segment s: A = ap[s], B = bp[s], both 256-wide
// panels): C -= sum_s A_s B_s^T in one pass of depth NSEG * KDEPTH.  Used by the single-GPU trailing update
// with NSEG = 4 (depth 1024 after a group of four panels): per-tile fixed costs (C read + write, pipeline fill)
// are 6.8 % of a depth-512 launch and 3.5 % of a depth-1024 one.
template <int NSEG>
struct SegPtrs {
    const double *a[NSEG];
    const double *b[NSEG];
};

// (Round 4 A/B: B staged in 32-deep chunks -- half the workgroup barriers, staging split in two halves so that no register is
// added, LDS 69.6 KB -- ran 66.4 against 69.3 TF in situ at N = 65 536, profiles/r04_kb32_ab.txt: not adopted, git history.)
// MT = 16-row m-tiles per wave: 2 -> 128 x 128 per workgroup, 1 -> 64 x 128 (half the time per tile: the last round of a short launch)
template <int NW, int KDEPTH, int NSEG, int MT = 2>
__device__ __forceinline__ void gemm_tile_dtv_segs(const SegPtrs<NSEG> &sp, double *c_ptr) {
    static_assert(NSEG >= 1, "compile-time segment list");
    constexpr int LD = TGP_PW;
    constexpr int LSB = DTV_LSB;
    constexpr int BPT = 16 / NW;
    constexpr int BROWS = 8 * NW;
    double (*ldsB)[128 * LSB] = reinterpret_cast<double (*)[128 * LSB]>(dtv_lds_storage());

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, l4 = lane >> 4;

    __amdgpu_buffer_rsrc_t ra[NSEG], rb[NSEG];
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        ra[s] = tile_rsrc(sp.a[s], 16 * MT * NW * LD * 8);
        rb[s] = tile_rsrc(sp.b[s], 128 * LD * 8);
    }
    const __amdgpu_buffer_rsrc_t rc = tile_rsrc(c_ptr, 16 * MT * NW * LD * 8);
    const int va = ((16 * MT * w + l15) * LD + 2 * l4) * 8;
    const int srow = tid >> 3, kp = (tid & 7) * 2;
    const int vb = (srow * LD + kp) * 8;
    const int vc = ((16 * MT * w + l4) * LD + l15) * 8;
    const int fb = l15 * LSB + 2 * l4;

    double2 areg[2][MT][2];                                     // [set][m][h]
    double2 rbst[BPT];
    auto load_a = [&](double2 (&dst)[MT][2], __amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) dst[m][h] = buf_ld2(src, va, (m * 16 * LD + k0 + 8 * h) * 8);
    };
    auto load_b = [&](__amdgpu_buffer_rsrc_t src, int k0) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) rbst[s] = buf_ld2(src, vb, (s * BROWS * LD + k0) * 8);
    };
    auto store_b = [&](int buf) {
#pragma unroll
        for (int s = 0; s < BPT; ++s) *reinterpret_cast<double2 *>(&ldsB[buf][(srow + BROWS * s) * LSB + kp]) = rbst[s];
    };
    load_a(areg[0], ra[0], 0);
    load_b(rb[0], 0);

    d4 acc[MT][8];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[m][n][r] = buf_ld1_stream(rc, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8);
    store_b(0);
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[m][n] = -acc[m][n];

    constexpr int cps = KDEPTH / KB;
    static_assert(cps % 2 == 0, "chunks are processed in register-set pairs");
    // chunk cc of a segment; `nsrc_*` = where the chunk after it comes from (same segment, or chunk 0 of the next)
    auto step = [&](const int cc, const bool last_seg, __amdgpu_buffer_rsrc_t sa, __amdgpu_buffer_rsrc_t sb,
                    __amdgpu_buffer_rsrc_t na, __amdgpu_buffer_rsrc_t nb, double2 (&cur)[MT][2], double2 (&nxt)[MT][2]) {
        const int buf = cc & 1;
        const bool wrap = (cc + 1 == cps);                       // wave-uniform
        const bool more = !(wrap && last_seg);
        if (more) {
            const int k0 = wrap ? 0 : (cc + 1) * KB;
            load_a(nxt, wrap ? na : sa, k0);
            load_b(wrap ? nb : sb, k0);
        }
        const double *Bs = ldsB[buf];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double2 bf[8];
#pragma unroll
            for (int n = 0; n < 8; ++n) bf[n] = *reinterpret_cast<const double2 *>(&Bs[fb + n * 16 * LSB + 8 * h]);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].x, bf[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(cur[m][h].y, bf[n].y, acc[m][n], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) store_b(buf ^ 1);
        __syncthreads();
    };
#pragma unroll
    for (int s = 0; s < NSEG; ++s) {
        const bool last = (s == NSEG - 1);
        const __amdgpu_buffer_rsrc_t na = ra[last ? s : s + 1], nb = rb[last ? s : s + 1];
#pragma unroll 1
        for (int cc = 0; cc < cps; cc += 2) {
            step(cc, last, ra[s], rb[s], na, nb, areg[0], areg[1]);
            step(cc + 1, last, ra[s], rb[s], na, nb, areg[1], areg[0]);
        }
    }

#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 8; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) buf_st1_stream(-acc[m][n][r], rc, vc, ((m * 16 + 4 * r) * LD + n * 16) * 8);
}
