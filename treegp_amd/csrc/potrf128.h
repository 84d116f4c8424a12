// potrf128: Cholesky factor L (in place, lower) and W = L^-1 of one 128x128 diagonal block, one
// workgroup of 4 waves, everything in LDS + registers.  This kernel is the serial critical path of
// every panel (and, in the multi-GPU solve, of every rank), so it is built for latency:
//
//   phase 1, per 32-column block jb:
//     wave 0   32x32 diagonal block -> L_jj and D_j = L_jj^-1, no workgroup barrier:
//              DIAG16 (default): two 16x16 halves, each by Gauss-Jordan in REGISTERS of one 16-lane DPP
//              row (lane i = row i; every broadcast is a `row_newbcast` DPP move, no SGPR round trip,
//              pivots through v_rsq_f64 + two Newton steps), glued by five 16x16x16 MFMA products
//              through LDS inside the wave (L21 = A21 D11^T, A22 -= L21 L21^T, D21 = -D22 L21 D11);
//              !DIAG16: the whole 32x32 block in registers with v_readlane broadcasts (3x slower)
//     4 waves  rows below:   X = A D_j^T           (v_mfma_f64_16x16x4_f64, operands from LDS)
//     4 waves  Schur update: T[r1:, r1:] -= X X^T  (lower 16x16 tiles)
//   phase 2: blocked in-place inversion of the 4x4 block-lower matrix (W_ii = D_i,
//            W_ij = -D_i sum_k L_ik W_kj), again 16x16x4 MFMA tiles from LDS.
//
// Explicit inverses only of well-conditioned diagonal blocks: cond(L_block) <= sqrt(cond(K)).
#pragma once
#include <utility>
#include "tgp_internal.h"

#ifdef TGP_POTRF_STAMPS
// one row of stamps per diagonal block of the factorisation (row = base / 128), so that every call can be looked at in situ
__device__ unsigned long long tgp_potrf_stamps[1024 * 20];
#define POTRF_STAMP(i) do { if (threadIdx.x == 0) tgp_potrf_stamps[((base >> 7) & 1023) * 20 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
// inside the diagonal step of 32-column block 1 (wave 0): shader-clock stamps between its sub-steps
__device__ unsigned long long tgp_potrf_fine[1024 * 8];
#define POTRF_FINE(k) do { if (jb == 1 && threadIdx.x == 0) tgp_potrf_fine[((base >> 7) & 1023) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define POTRF_FINE(k) do { } while (0)
#define POTRF_STAMP(i) do { } while (0)
#endif

namespace potrf_v2 {
typedef double d4v __attribute__((ext_vector_type(4)));
// LDS image: only the 10 lower 32x32 blocks of the 4x4 block matrix, each with row stride 34 doubles
// (fragment reads: lane (r, kq) -> slot 34 r + kq, conflict-free per 32-lane group).  87 KB, so the
// kernel fits beside one 70 KB GEMM workgroup on a CU -- that is what lets the look-ahead side stream
// get a CU while the trailing update saturates the chip.
constexpr int BS = 34;                    // row stride inside a block
constexpr int BLK_ELEMS = 32 * BS;
constexpr int T_ELEMS = 10 * BLK_ELEMS;
__device__ __forceinline__ int taddr(int r, int c) {           // (r, c) with block(r) >= block(c)
    const int bi = r >> 5, bj = c >> 5;
    return (bi * (bi + 1) / 2 + bj) * BLK_ELEMS + (r & 31) * BS + (c & 31);
}

__device__ __forceinline__ double readlane_f64(double x, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}

// In-register Gauss-Jordan on a 32x32 SPD block: lane i holds row i in t[0..31].
// On exit t[c] = (L^-1)[i][c] and ls[c] = L[i][c] for c <= i.  fail = first bad pivot or -1.
__device__ __forceinline__ void gauss_jordan32(double (&t)[32], double (&ls)[32], const int i, int &fail) {
#pragma clang loop unroll(full)
    for (int j = 0; j < 32; ++j) {
        const double pj = readlane_f64(t[j], j);
        if (!(pj > 0.0) && fail < 0) fail = j;
        const double d = sqrt(pj);
        const double inv = 1.0 / d;
        const double li = t[j] * inv;                       // L[i][j] for i > j
        if (i == j) {                                       // row j of L^-1: scale the pivot row
#pragma clang loop unroll(full)
            for (int c = 0; c < j; ++c) t[c] *= inv;
            t[j] = inv;
        }
        ls[j] = (i > j) ? li : ((i == j) ? d : 0.0);
        if (i > j) {
#pragma clang loop unroll(full)
            for (int c = 0; c < 32; ++c) {
                double vc;
                if (c > j) vc = readlane_f64(li, c);        // L[c][j]
                else if (c < j) vc = readlane_f64(t[c], j); // (L^-1)[j][c]
                else vc = inv;
                t[c] = ((c == j) ? 0.0 : t[c]) - li * vc;
            }
        }
    }
}

// ---- 16x16 Gauss-Jordan inside one DPP row -------------------------------------------------------------
// lane N of every 16-lane row to all lanes of that row (gfx90a+ DPP control row_newbcast:N)
template <int N>
__device__ __forceinline__ double bcast16(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x150 + N, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x150 + N, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// 1/sqrt(p) to ~1 ulp: hardware estimate + two Newton steps (the pivot chain is the critical path).  Measured with
// tools/probes/gj16_probe.hip: v_rsq_f64 alone leaves 6e-8 in L, one step 2.2e-15 (five times the two-step error, for 5 % of a
// sweep): a backward error ten times larger is ten times the forward error at cond(K) ~ 1e6, where 1e-10 is the budget.
#ifndef TGP_RSQ_NR
#define TGP_RSQ_NR 2
#endif
__device__ __forceinline__ double rsqrt_nr(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double hp = 0.5 * p;
#pragma unroll
    for (int it = 0; it < TGP_RSQ_NR; ++it) {
        const double e = __builtin_fma(-hp * y, y, 0.5);
        y = __builtin_fma(y, e, y);
    }
    return y;
}
#include "gj16_dpp.h"
template <int J, int C>
__device__ __forceinline__ void gj16_elem(double (&t)[16], const double li, const double lm) {
    if constexpr (C > J) t[C] = __builtin_fma(-lm, bcast16<C>(li), t[C]);            // L[C][J] lives in lane C
    else if constexpr (C < J) t[C] = __builtin_fma(-lm, bcast16<J>(t[C]), t[C]);     // (L^-1)[J][C]: pivot row, lane J
}
template <int J, int... Cs>
__device__ __forceinline__ void gj16_column(double (&t)[16], double (&ls)[16], const int i, int &fail,
                                            std::integer_sequence<int, Cs...>) {
    const double pj = bcast16<J>(t[J]);
    if (!(pj > 0.0) && fail < 0) fail = J;
    const double inv = rsqrt_nr(pj);
    const double d = pj * inv;
    const double li = t[J] * inv;                         // L[i][J] for the rows below the pivot
    const double scale = (i == J) ? inv : 1.0;            // row J of L^-1: scale the pivot row
    ((t[Cs] = (Cs < J) ? t[Cs] * scale : t[Cs]), ...);
    const double lm = (i > J) ? li : 0.0;                 // rows <= J are finished: multiplier 0
#ifdef TGP_GJ16_PORTABLE
    (gj16_elem<J, Cs>(t, li, lm), ...);                   // two DPP moves + one FMA per element
#else
    gj16_update<J>(t, li, -lm);                           // one v_fmac_f64_dpp per element (gj16_dpp.h)
#endif
    t[J] = (i > J) ? -li * inv : ((i == J) ? inv : t[J]);
    ls[J] = (i > J) ? li : ((i == J) ? d : 0.0);
}
template <int... Js>
__device__ __forceinline__ void gj16_all(double (&t)[16], double (&ls)[16], const int i, int &fail,
                                         std::integer_sequence<int, Js...> seq) {
    (gj16_column<Js>(t, ls, i, fail, seq), ...);
}
// lanes 0..15 of a wave (lane i = row i): on exit t[c] = (L^-1)[i][c], ls[c] = L[i][c] for c <= i
__device__ __forceinline__ void gauss_jordan16(double (&t)[16], double (&ls)[16], const int i, int &fail) {
    gj16_all(t, ls, i, fail, std::make_integer_sequence<int, 16>{});
}

// ---- the same over all four DPP rows of the wave (gen_gj16s.py) -------------------------------------------------------------
// lane (r, i): s[] = row i of the Schur / Cholesky columns (replicated in the four rows), w[k] = column 4 k + r of the inverse,
// unscaled until the end.  On exit w[k] = (L^-1)[i][4 k + r], ls[c] = L[i][c] (every DPP row holds all of ls).
#include "gj16s_dpp.h"
template <int J>
__device__ __forceinline__ void gj16s_column(double (&s)[16], double (&w)[4], double (&ls)[16], double &myinv, const int i, const int r,
                                             int &fail) {
    const double pj = bcast16<J>(s[J]);
    if (!(pj > 0.0) && fail < 0) fail = J;
    const double inv = rsqrt_nr(pj);
    const double d = pj * inv;
    const double li = s[J] * inv;                         // L[i][J] for the rows below the pivot
    const double nl = (i > J) ? -li : 0.0;                // rows <= J are finished: multiplier 0
    const double nl2 = nl * inv;                          // times (L^-1)[J][C] = inv * (row J's unscaled entry)
    const double nl2m = (r < (J & 3)) ? nl2 : 0.0;        // slot J / 4 holds a column < J only in the DPP rows r < J % 4
    gj16s_update<J>(s, w, li, nl, nl2, nl2m);
    // column J of the inverse is born in DPP row J % 4, slot J / 4: -L[i][J] / L[J][J]^2 below the diagonal, 1 (unscaled) on it
    const double born = (i == J) ? 1.0 : nl2;
    w[J >> 2] = (r == (J & 3)) ? born : w[J >> 2];
    myinv = (i == J) ? inv : myinv;
    ls[J] = (i > J) ? li : ((i == J) ? d : 0.0);
}
template <int... Js>
__device__ __forceinline__ void gj16s_all(double (&s)[16], double (&w)[4], double (&ls)[16], double &myinv, const int i, const int r,
                                          int &fail, std::integer_sequence<int, Js...>) {
    (gj16s_column<Js>(s, w, ls, myinv, i, r, fail), ...);
}
__device__ __forceinline__ void gauss_jordan16s(double (&s)[16], double (&w)[4], double (&ls)[16], const int i, const int r, int &fail) {
    double myinv = 1.0;
    gj16s_all(s, w, ls, myinv, i, r, fail, std::make_integer_sequence<int, 16>{});
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; ++k) w[k] *= myinv;
}

// ---- third form (round 5, gen_gj16t.py): the per-pivot bookkeeping taken out of the instruction stream -------------------------
// The sweep is issue-bound (tools/probes/issue_probe.hip: 4.5 clocks per 64-bit VALU instruction dependent or not, 16.5 for
// v_rsq_f64, 12.5 for a DPP read of a fresh result), so what counts is the number of instructions per pivot: 37 in the second
// form, of which 10.4 are the updates.  Here: every lane takes the reciprocal square root of its OWN s[J] (lane J's is the pivot's;
// one third-order step instead of two Newton steps), one v_mov_b64_dpp broadcasts it, finished rows are not masked out of the Schur
// update (nobody reads what they compute), the inverse starts as the identity so that its columns are born by the ordinary
// update, the failure check is one look at the reciprocals at the end: 26 instructions per pivot.
#include "gj16t_dpp.h"
// 1/sqrt(p): hardware estimate y0 (relative error e0 ~ 6e-8), then y = y0 (1 + e + 1.5 e^2) with e = 0.5 - 0.5 p y0^2: error O(e0^3)
__device__ __forceinline__ double rsqrt_cubic(double p) {
    const double y0 = __builtin_amdgcn_rsq(p);
    const double e = __builtin_fma(-0.5 * p * y0, y0, 0.5);
    return __builtin_fma(y0 * e, __builtin_fma(1.5, e, 1.0), y0);
}
template <int J>
__device__ __forceinline__ void gj16t_column(double (&s)[16], double (&w)[4], double (&ls)[16], double &myinv, const int i) {
    const double y = rsqrt_cubic(s[J]);                   // lane J: 1 / L[J][J]
    const double yb = gj16t_bcast<J>(y);
    const double li = s[J] * yb;                          // L[i][J] for i >= J (lane J: the pivot's square root)
    const double nl2 = (i > J) ? -(li * yb) : 0.0;        // finished rows of the inverse stay
    gj16t_update<J>(s, w, li, nl2);
    // lane i keeps the last y it sees here, which is step i's: one family of lane masks (i > J), J = -1 .. 15, serves all three selects
    // (with (i == J) as well the masks of a sweep no longer fit in the scalar registers, and their spills take VGPRs from the top)
    myinv = (i >= J) ? y : myinv;
    ls[J] = (i >= J) ? li : 0.0;
}
template <int... Js>
__device__ __forceinline__ void gj16t_all(double (&s)[16], double (&w)[4], double (&ls)[16], double &myinv, const int i,
                                          std::integer_sequence<int, Js...>) {
    (gj16t_column<Js>(s, w, ls, myinv, i), ...);
}
// lane (r, i): s[] = row i of the block (what lies above the diagonal does not matter).  On exit w[k] = (L^-1)[i][4 k + r] (exact
// zeros above the diagonal), ls[c] = L[i][c] (zeros above the diagonal); fail = first non-positive (or NaN) pivot, or stays < 0.
__device__ __forceinline__ void gauss_jordan16t(double (&s)[16], double (&w)[4], double (&ls)[16], const int i, const int r, int &fail) {
    double myinv = 1.0;
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; ++k) w[k] = (4 * k + r == i) ? 1.0 : 0.0;
    gj16t_all(s, w, ls, myinv, i, std::make_integer_sequence<int, 16>{});
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; ++k) w[k] *= myinv;
    // a pivot <= 0 (or NaN) leaves NaN (or inf) in its reciprocal and in every later one: the first such lane is the failing pivot
    const unsigned long long bad = __ballot(!(myinv > 0.0 && myinv < __builtin_huge_val())) & 0xffffull;
    if (bad != 0ull && fail < 0) fail = __builtin_ctzll(bad);
}

// acc += A[ra.., ca..ca+31] (16 x 32) * B[rb.., cb..cb+31]^T (16 x 32), both row-major in T
// (splitting these products over two accumulator chains was tried and is slower: the extra adds and LDS
//  traffic cost more than the dependent-MFMA latency they hide)
__device__ __forceinline__ d4v mma_nt32(const double *T, int ra, int ca, int rb, int cb, d4v acc, int l15, int l4) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double a = T[taddr(ra + l15, ca + 4 * ks + l4)];
        const double b = T[taddr(rb + l15, cb + 4 * ks + l4)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}
// acc += A[ra.., ca..ca+31] (16 x 32) * B[rb..rb+31, cb..] (32 x 16)
__device__ __forceinline__ d4v mma_nn32(const double *T, int ra, int ca, int rb, int cb, d4v acc, int l15, int l4) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const double a = T[taddr(ra + l15, ca + 4 * ks + l4)];
        const double b = T[taddr(rb + 4 * ks + l4, cb + l15)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// 16-deep products on the block-packed image (operands inside one 32x32 block)
__device__ __forceinline__ d4v mma_nt16(const double *T, int ra, int ca, int rb, int cb, int l15, int l4) {
    d4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[taddr(ra + l15, ca + 4 * ks + l4)], T[taddr(rb + l15, cb + 4 * ks + l4)], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ d4v mma_nn16(const double *T, int ra, int ca, int rb, int cb, int l15, int l4) {
    d4v acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T[taddr(ra + l15, ca + 4 * ks + l4)], T[taddr(rb + 4 * ks + l4, cb + l15)], acc, 0, 0, 0);
    return acc;
}

// one 16x16 diagonal half at (q, q): lanes 0..15 factor it in registers; D replaces it in T, L goes to A
__device__ __forceinline__ void diag16(double *T, double *A, int lda, int q, int lane, int *info, int base) {
    if (lane < 16) {
        const int i = lane;
        double t[16], ls[16];
#pragma clang loop unroll(full)
        for (int c = 0; c < 16; ++c) {
            const double v = T[taddr(q + i, q + c)];
            t[c] = (c <= i) ? v : 0.0;
        }
        int fail = -1;
        gauss_jordan16(t, ls, i, fail);
        if (fail >= 0 && i == 0) atomicCAS(info, 0, base + q + fail + 1);
#pragma clang loop unroll(full)
        for (int c = 0; c < 16; ++c) {
            T[taddr(q + i, q + c)] = (c <= i) ? t[c] : 0.0;
            if (c <= i) A[(int64_t)(q + i) * lda + q + c] = ls[c];
        }
    }
}

// the same with the Gauss-Jordan sweep laid out over the four DPP rows of the wave (gauss_jordan16s): 3333 against 5026 clock ticks
// per block for load + sweep + store on an otherwise idle chip (tools/probes/gj16_probe.hip)
// blk: the 16x16 block inside its 32x32 LDS block (row stride BS); Ag: the same block in global memory (row stride lda); pos: its
// global row / column for the failure report.  All addresses are one base plus compile-time offsets: with taddr() per element
// the sweep cost 4800 clock ticks in situ against the probe's 3333, the difference being integer address arithmetic and
// sixteen separately predicated global stores (the upper triangle of ls is zero anyway: it is stored as such).
__device__ __forceinline__ void diag16s(double *blk, double *Ag, int lda, int lane, int *info, int pos) {
    const int i = lane & 15, r = lane >> 4;
    double s[16], w[4] = {0.0, 0.0, 0.0, 0.0}, ls[16];
    const double *row = blk + i * BS;
#pragma clang loop unroll(full)
    for (int c = 0; c < 16; ++c) s[c] = row[c];
#pragma clang loop unroll(full)
    for (int c = 0; c < 16; ++c) s[c] = (c <= i) ? s[c] : 0.0;
    int fail = -1;
    gauss_jordan16s(s, w, ls, i, r, fail);
    if (fail >= 0 && lane == 0) atomicCAS(info, 0, pos + fail + 1);
    double *wrow = blk + i * BS + r;                                     // D replaces the block: every DPP row its own columns
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; ++k) wrow[4 * k] = (4 * k + r <= i) ? w[k] : 0.0;
    if (r == 0) {                                                        // L is final (zeros above the diagonal of the block)
        double *grow = Ag + (int64_t)i * lda;
#pragma clang loop unroll(full)
        for (int c = 0; c < 16; ++c) grow[c] = ls[c];
    }
}
// third form of the sweep: 16-byte LDS reads, nothing masked on the way in or out (gauss_jordan16t)
__device__ __forceinline__ void diag16t(double *blk, double *Ag, int lda, int lane, int *info, int pos) {
    const int i = lane & 15, r = lane >> 4;
    double s[16], w[4], ls[16];
    const double2 *row = reinterpret_cast<const double2 *>(blk + i * BS);      // BS even and the block's origin even: 16-byte aligned
#pragma clang loop unroll(full)
    for (int c = 0; c < 8; ++c) {
        const double2 v = row[c];
        s[2 * c] = v.x;
        s[2 * c + 1] = v.y;
    }
    int fail = -1;
    gauss_jordan16t(s, w, ls, i, r, fail);
    if (fail >= 0 && lane == 0) atomicCAS(info, 0, pos + fail + 1);
    double *wrow = blk + i * BS + r;                                     // D replaces the block: every DPP row its own columns
#pragma clang loop unroll(full)
    for (int k = 0; k < 4; ++k) wrow[4 * k] = w[k];
    if (r == 0) {                                                        // L is final (zeros above the diagonal of the block)
        double *grow = Ag + (int64_t)i * lda;
#pragma clang loop unroll(full)
        for (int c = 0; c < 16; ++c) grow[c] = ls[c];
    }
}
// 16-deep products with both operands inside one 32x32 LDS block B (row stride BS): offsets are compile-time
template <int RA, int CA, int RB, int CB>
__device__ __forceinline__ d4v blk_nt16(const double *B, int l15, int l4) {      // A[RA.., CA..CA+15] * B[RB.., CB..CB+15]^T
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    const double *a = B + (RA + l15) * BS + CA + l4, *b = B + (RB + l15) * BS + CB + l4;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], b[4 * ks], acc, 0, 0, 0);
    return acc;
}
template <int RA, int CA, int RB, int CB>
__device__ __forceinline__ d4v blk_nn16(const double *B, int l15, int l4) {      // A[RA.., CA..CA+15] * B[RB..RB+15, CB..]
    d4v acc = {0.0, 0.0, 0.0, 0.0};
    const double *a = B + (RA + l15) * BS + CA + l4, *b = B + (RB + l4) * BS + CB + l15;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], b[4 * ks * BS], acc, 0, 0, 0);
    return acc;
}

// phase 2 of potrf128, one off-diagonal 32x32 block of W = L^-1:  W_IB,CB = -D_IB sum_{k = CB}^{IB-1} L_IB,k W_k,CB, this wave's
// 16x16 tile (rt, ct) of it.  The sum goes to the spare block SC (all four waves), then D_IB SC replaces L_IB,CB in place.
__device__ __forceinline__ constexpr int blk_origin(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * BLK_ELEMS; }
__device__ __forceinline__ constexpr int blk_row(int b) { return b < 1 ? 0 : (b < 3 ? 1 : (b < 6 ? 2 : 3)); }      // block row of lower block b
template <int IB, int CB>
__device__ __forceinline__ void phase2_pair(double *T, double *SC, int rt, int ct, int l15, int l4) {
    const int aoff = (16 * rt + l15) * BS + l4;        // A fragment: row 16 rt + l15, column 4 ks + l4 of a block
    const int boff = l4 * BS + 16 * ct + l15;          // B fragment: row 4 ks + l4, column 16 ct + l15
    const int coff = (16 * rt + l4) * BS + 16 * ct + l15;      // C fragment: rows 16 rt + l4 + 4 r
    d4v s = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kb = CB; kb < IB; ++kb) {
        const double *a = T + blk_origin(IB, kb) + aoff, *b = T + blk_origin(kb, CB) + boff;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) s = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], b[4 * ks * BS], s, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) SC[coff + 4 * r * BS] = s[r];
    __syncthreads();
    d4v w = {0.0, 0.0, 0.0, 0.0};
    const double *a = T + blk_origin(IB, IB) + aoff, *b = SC + boff;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], b[4 * ks * BS], w, 0, 0, 0);
    double *c = T + blk_origin(IB, CB) + coff;
#pragma unroll
    for (int r = 0; r < 4; ++r) c[4 * r * BS] = -w[r];
    __syncthreads();
}

// ---- the inversion under the diagonal steps (round 5) ---------------------------------------------------------------------------
// W = L^-1 used to be a phase of its own after the factorisation: six (block row, block column) pairs, two workgroup barriers each,
// 6.2 of the kernel's 28 us -- while during the four diagonal steps (13.8 us, wave 0 alone) three waves and their MFMA pipes had
// nothing to do.  Now the chains of the inverse's block columns
//     S_IB = sum_{k = CB}^{IB-1} L_IB,k W_k,CB,     W_IB,CB = -D_IB S_IB          (W_CB,CB = D_CB)
// run on the waves 1..3 UNDER the diagonal steps, in 16-column slices (a column of the inverse depends on nothing but itself): waves
// 1 and 2 own the two slices of block columns 0 and 2, wave 3 block column 1.  S and then W replace L_IB,CB in the LDS image (the
// image and the final write-out of W stay as they were) once that block of L is in global memory and everybody has read it:
//   under step 1 (D0)   w1, w2: S10, S20, S30 = L_i0 D0 (slices, in registers until w3 has written L_i0 out and all have read it)
//   under step 2 (D1)   w1, w2: W10 = -D1 S10, S20 += L21 W10, S30 += L31 W10;  w3: S21 = L21 D1 -> spare block, S31 = L31 D1
//   under step 3 (D2)   w1, w2: W20 = -D2 S20, S30 += L32 W20, S32 = L32 D2;    w3: W21 = -D2 S21, S31 += L32 W21
//   after step 3 (D3)   W30, W31, W32 = -D3 S3x: at most two half products per wave (wave 0 takes one slice of W31)
// What crosses waves goes through the workgroup barriers of the factorisation loop, except "I have read this block" inside a step:
// those are flags in LDS (release / acquire at workgroup scope).  Every wait is bounded -- all four waves are resident by
// construction, the bound only turns a logic error into info = -8 instead of a hang.
// acc[rt] += A (32 x 32 at Ablk) B[:, 16 h .. 16 h + 15] (B 32 x 32 at Bblk): one wave, a 32 x 16 slice, k = 4 ks + l4.
// A_LOWER / B_LOWER: that operand is a lower-triangular diagonal block -- products with its zero upper-right tile are skipped.
template <bool A_LOWER, bool B_LOWER>
__device__ __forceinline__ void slice_mm(const double *Ablk, const double *Bblk, int h, d4v (&acc)[2], int l15, int l4) {
    const double *a = Ablk + l15 * BS + l4, *b = Bblk + l4 * BS + 16 * h + l15;
    // (the scheduler would hoist every operand read of every product of a step to its top: 330 VGPRs, where the kernel has to stay
    //  within 264 to fit on a SIMD beside a wave of the trailing update -- hence the fences)
    __builtin_amdgcn_sched_barrier(0);
    if (B_LOWER && h == 1) {                             // rows 0..15 of those columns of B are zero: k from 16 on
#pragma unroll
        for (int ks = 4; ks < 8; ++ks) {
            const double bv = b[4 * ks * BS];
            if (!A_LOWER) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[16 * BS + 4 * ks], bv, acc[1], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const double bv = b[4 * ks * BS];
            if (!(A_LOWER && ks >= 4)) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], bv, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[16 * BS + 4 * ks], bv, acc[1], 0, 0, 0);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void slice_zero(d4v (&acc)[2]) {
    acc[0] = (d4v){0.0, 0.0, 0.0, 0.0};
    acc[1] = (d4v){0.0, 0.0, 0.0, 0.0};
}
__device__ __forceinline__ void slice_load(const double *blk, int h, d4v (&acc)[2], int l15, int l4) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[rt][r] = blk[(16 * rt + l4 + 4 * r) * BS + 16 * h + l15];
}
template <bool NEGATE>
__device__ __forceinline__ void slice_store(double *blk, int h, const d4v (&acc)[2], int l15, int l4) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) blk[(16 * rt + l4 + 4 * r) * BS + 16 * h + l15] = NEGATE ? -acc[rt][r] : acc[rt][r];
    __builtin_amdgcn_sched_barrier(0);
}
// one 32 x 32 block of L from the LDS image to global memory, one wave: lane = (row, half row)
__device__ __forceinline__ void wave_block_to_global(const double *blk, double *Ag, int lda, int lane) {
    const int row = lane >> 1, half = lane & 1;
    const double2 *src = reinterpret_cast<const double2 *>(blk + row * BS + 16 * half);
    double2 *dst = reinterpret_cast<double2 *>(Ag + (int64_t)row * lda + 16 * half);
    double2 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = src[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) dst[q] = v[q];
    __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void helper_post(int *flag) {
    __hip_atomic_store(flag, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void helper_wait(int *flag, int *info) {
    unsigned it = 0;
    while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++it > (1u << 22)) {                          // cannot happen: the poster is a resident wave of this workgroup
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(info, -8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}
// flag words: R0[1..3] "wave w no longer reads L_i0" (step 1), R1[1..2] "... L31" (step 2), R2[1..3] "... L32" (step 3)
constexpr int HF_R0 = 0, HF_R1 = 4, HF_R2 = 8, HF_WORDS = 12;

// One step of the schedule above for the waves 1..3 (jb = the factorisation's iteration that has just produced D_jb and L_i,jb).
__device__ __forceinline__ void inversion_step(double *T, double *A, int lda, int *info, int jb, int wave, int lane) {
    const int l15 = lane & 15, l4 = lane >> 4;
    double *SC = T + T_ELEMS;
    int *hflag = reinterpret_cast<int *>(T + T_ELEMS + BLK_ELEMS);
    const int h = wave - 1;                            // waves 1, 2: their 16-column slice of block columns 0 and 2
    double *const B10 = T + blk_origin(1, 0), *const B20 = T + blk_origin(2, 0), *const B30 = T + blk_origin(3, 0);
    double *const B21 = T + blk_origin(2, 1), *const B31 = T + blk_origin(3, 1), *const B32 = T + blk_origin(3, 2);
    const double *const D0 = T + blk_origin(0, 0), *const D1 = T + blk_origin(1, 1), *const D2 = T + blk_origin(2, 2);
    if (jb == 0) {
        if (wave == 3) {                               // L_i0 to global memory, then the slices may replace it; S30 (both slices)
            wave_block_to_global(B10, A + (int64_t)32 * lda, lda, lane);
            wave_block_to_global(B20, A + (int64_t)64 * lda, lda, lane);
            wave_block_to_global(B30, A + (int64_t)96 * lda, lda, lane);
            d4v t0[2], t1[2];
            slice_zero(t0); slice_zero(t1);
            slice_mm<false, true>(B30, D0, 0, t0, l15, l4);            // S30 = L30 D0
            slice_mm<false, true>(B30, D0, 1, t1, l15, l4);
            helper_post(hflag + HF_R0 + 3);
            slice_store<false>(B30, 0, t0, l15, l4);                   // (waves 1 and 2 do not read L30 in this step)
            slice_store<false>(B30, 1, t1, l15, l4);
        } else {
            d4v s1[2], s2[2];
            slice_zero(s1); slice_zero(s2);
            slice_mm<false, true>(B10, D0, h, s1, l15, l4);            // S_i0 = L_i0 D0
            slice_mm<false, true>(B20, D0, h, s2, l15, l4);
            helper_post(hflag + HF_R0 + wave);
            helper_wait(hflag + HF_R0 + (3 - wave), info);
            helper_wait(hflag + HF_R0 + 3, info);
            slice_store<false>(B10, h, s1, l15, l4);
            slice_store<false>(B20, h, s2, l15, l4);
        }
    } else if (jb == 1) {
        if (wave == 3) {
            d4v s[2], t0[2], t1[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                           // S21 = L21 D1 -> spare block
                slice_zero(s);
                slice_mm<false, true>(B21, D1, hh, s, l15, l4);
                slice_store<false>(SC, hh, s, l15, l4);
            }
            slice_zero(t0); slice_zero(t1);
            slice_mm<false, true>(B31, D1, 0, t0, l15, l4);            // S31 = L31 D1 (the L32 W21 term under the next step)
            slice_mm<false, true>(B31, D1, 1, t1, l15, l4);
            helper_wait(hflag + HF_R1 + 1, info);                      // waves 1 and 2 have read L31 (and written it out)
            helper_wait(hflag + HF_R1 + 2, info);
            slice_store<false>(B31, 0, t0, l15, l4);
            slice_store<false>(B31, 1, t1, l15, l4);
        } else {
            if (wave == 1) wave_block_to_global(B21, A + (int64_t)64 * lda + 32, lda, lane);
            else wave_block_to_global(B31, A + (int64_t)96 * lda + 32, lda, lane);
            d4v acc[2];
            slice_zero(acc);
            slice_mm<true, false>(D1, B10, h, acc, l15, l4);           // W10 = -D1 S10
            slice_store<true>(B10, h, acc, l15, l4);
            slice_load(B20, h, acc, l15, l4);
            slice_mm<false, false>(B21, B10, h, acc, l15, l4);         // S20 += L21 W10
            slice_store<false>(B20, h, acc, l15, l4);
            slice_load(B30, h, acc, l15, l4);
            slice_mm<false, false>(B31, B10, h, acc, l15, l4);         // S30 += L31 W10
            slice_store<false>(B30, h, acc, l15, l4);
            helper_post(hflag + HF_R1 + wave);
        }
    } else if (jb == 2) {
        if (wave == 3) {
            d4v acc[2];
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                           // W21 = -D2 S21 (L21 is in global memory since step 2)
                slice_zero(acc);
                slice_mm<true, false>(D2, SC, hh, acc, l15, l4);
                slice_store<true>(B21, hh, acc, l15, l4);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {                           // S31 += L32 W21
                slice_load(B31, hh, acc, l15, l4);
                slice_mm<false, false>(B32, B21, hh, acc, l15, l4);
                slice_store<false>(B31, hh, acc, l15, l4);
            }
            helper_post(hflag + HF_R2 + 3);
        } else {
            if (wave == 1) wave_block_to_global(B32, A + (int64_t)96 * lda + 64, lda, lane);
            d4v acc[2], s32[2];
            slice_zero(acc);
            slice_mm<true, false>(D2, B20, h, acc, l15, l4);           // W20 = -D2 S20
            slice_store<true>(B20, h, acc, l15, l4);
            slice_load(B30, h, acc, l15, l4);
            slice_mm<false, false>(B32, B20, h, acc, l15, l4);         // S30 += L32 W20
            slice_store<false>(B30, h, acc, l15, l4);
            slice_zero(s32);
            slice_mm<false, true>(B32, D2, h, s32, l15, l4);           // S32 = L32 D2
            helper_post(hflag + HF_R2 + wave);
            helper_wait(hflag + HF_R2 + (3 - wave), info);
            helper_wait(hflag + HF_R2 + 3, info);
            slice_store<false>(B32, h, s32, l15, l4);
        }
    }
}

// Register budget: at most 264 VGPRs (arch + acc), so that a wave of this kernel fits on a SIMD next to one wave of
// the trailing update (248 of 512) -- with more it has to wait for an EMPTY compute unit during the look-ahead
// (measured: 268 VGPRs cost 25 ms of exposed panel time at N = 65536).  The initial load is batched in two halves
// for that reason; tools/check_potrf_regs.sh (run by the build) fails if the budget is exceeded.
// -DTGP_GJ16_ONE_ROW: A/B build with the sweep on the 16 lanes of one DPP row and taddr() addressing (round 1)
// The body works on an LDS image T of POTRF_LDS_DOUBLES doubles handed in by the kernel, so that a kernel that also runs GEMM tiles
// (chol.hip: panel_mid_kernel) can alias their staging over it.
constexpr int POTRF_LDS_DOUBLES = T_ELEMS + BLK_ELEMS + 8;     // 10 lower blocks + one spare block + the helper waves' flags (12 words)
template <bool DIAG16>
__device__ __forceinline__ void potrf128_body(double *T, double *A, int lda, double *W, int *info, int base) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const d4v zero4 = {0.0, 0.0, 0.0, 0.0};
    POTRF_STAMP(0);

    // block-wise copies between global memory and the LDS image: a pass of the 256 threads covers 16 rows x 16 column pairs = half a
    // 32x32 block, so block and half are compile-time and every address is one per-thread base plus a constant
    const int lrow = tid >> 4, lpair = tid & 15;
    double *tbase = T + lrow * BS + 2 * lpair;
    double *SC = T + T_ELEMS;                          // the spare 32x32 block
    int *hflag = reinterpret_cast<int *>(T + T_ELEMS + BLK_ELEMS);      // helper waves' "I have read it" events (see above)
    if (tid < HF_WORDS) hflag[tid] = 0;
    {   // the 10 lower blocks come in with all 20 loads of a thread in flight at once (one memory round trip), 16-byte LDS writes;
        // the diagonal blocks are masked to their lower triangles
        const double *gbase = A + (int64_t)lrow * lda + 2 * lpair;
        double2 v[20];
#pragma unroll
        for (int u = 0; u < 20; ++u) {
            const int bi = blk_row(u >> 1), bj = (u >> 1) - bi * (bi + 1) / 2, h = u & 1;
            v[u] = *reinterpret_cast<const double2 *>(gbase + (int64_t)(32 * bi + 16 * h) * lda + 32 * bj);
        }
#pragma unroll
        for (int u = 0; u < 20; ++u) {
            const int bi = blk_row(u >> 1), bj = (u >> 1) - bi * (bi + 1) / 2, h = u & 1;
            double2 x = v[u];
            if (bi == bj) {
                const int row = lrow + 16 * h, c = 2 * lpair;
                x.x = (c <= row) ? x.x : 0.0;
                x.y = (c + 1 <= row) ? x.y : 0.0;
            }
            *reinterpret_cast<double2 *>(tbase + (u >> 1) * BLK_ELEMS + 16 * h * BS) = x;
        }
    }
    __syncthreads();

    // ---- phase 1: factorisation ---------------------------------------------------------------
#pragma unroll 1
    for (int jb = 0; jb < 4; ++jb) {
        const int r0 = 32 * jb;
        POTRF_STAMP(1 + 3 * jb);
        if (DIAG16 && wave == 0) {
            // whole wave, no workgroup barrier: LDS operations of one wave execute in program order
            POTRF_FINE(0);
#ifdef TGP_GJ16_ONE_ROW
            diag16(T, A, lda, r0, lane, info, base);
            POTRF_FINE(1);
            const d4v x = mma_nt16(T, r0 + 16, r0, r0, r0, l15, l4);                   // L21 = A21 D11^T
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                T[taddr(r0 + 16 + l4 + 4 * r, r0 + l15)] = x[r];
                A[(int64_t)(r0 + 16 + l4 + 4 * r) * lda + r0 + l15] = x[r];
            }
            POTRF_FINE(2);
            const d4v p = mma_nt16(T, r0 + 16, r0, r0 + 16, r0, l15, l4);              // A22 -= L21 L21^T
#pragma unroll
            for (int r = 0; r < 4; ++r) T[taddr(r0 + 16 + l4 + 4 * r, r0 + 16 + l15)] -= p[r];
            POTRF_FINE(3);
            diag16(T, A, lda, r0 + 16, lane, info, base);
            POTRF_FINE(4);
            const d4v s = mma_nn16(T, r0 + 16, r0, r0, r0, l15, l4);                   // S = L21 D11
#pragma unroll
            for (int r = 0; r < 4; ++r) T[taddr(r0 + 16 + l4 + 4 * r, r0 + l15)] = s[r];
            POTRF_FINE(5);
            const d4v w = mma_nn16(T, r0 + 16, r0 + 16, r0 + 16, r0, l15, l4);         // D21 = -D22 S
#pragma unroll
            for (int r = 0; r < 4; ++r) T[taddr(r0 + 16 + l4 + 4 * r, r0 + l15)] = -w[r];
#else
            double *B = T + taddr(r0, r0);                      // the 32x32 diagonal block of this step, row stride BS
            double *Ag = A + (int64_t)r0 * lda + r0;            // the same block in global memory
#ifdef TGP_GJ16_SECOND
            diag16s(B, Ag, lda, lane, info, base + r0);
#else
            diag16t(B, Ag, lda, lane, info, base + r0);
#endif
            POTRF_FINE(1);
            double *b21 = B + (16 + l4) * BS + l15;             // C fragment of the lower-left 16x16 block: rows 16 + l4 + 4 r
            const d4v x = blk_nt16<16, 0, 0, 0>(B, l15, l4);                           // L21 = A21 D11^T
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                b21[4 * r * BS] = x[r];
                Ag[(int64_t)(16 + l4 + 4 * r) * lda + l15] = x[r];
            }
            POTRF_FINE(2);
            const d4v p = blk_nt16<16, 0, 16, 0>(B, l15, l4);                          // A22 -= L21 L21^T
#pragma unroll
            for (int r = 0; r < 4; ++r) b21[4 * r * BS + 16] -= p[r];
            POTRF_FINE(3);
#ifdef TGP_GJ16_SECOND
            diag16s(B + 16 * BS + 16, Ag + (int64_t)16 * lda + 16, lda, lane, info, base + r0 + 16);
#else
            diag16t(B + 16 * BS + 16, Ag + (int64_t)16 * lda + 16, lda, lane, info, base + r0 + 16);
#endif
            POTRF_FINE(4);
            const d4v s = blk_nn16<16, 0, 0, 0>(B, l15, l4);                           // S = L21 D11
#pragma unroll
            for (int r = 0; r < 4; ++r) b21[4 * r * BS] = s[r];
            POTRF_FINE(5);
            const d4v w = blk_nn16<16, 16, 16, 0>(B, l15, l4);                         // D21 = -D22 S
#pragma unroll
            for (int r = 0; r < 4; ++r) b21[4 * r * BS] = -w[r];
#endif
            POTRF_FINE(6);
        }
        if (!DIAG16 && wave == 0 && lane < 32) {
            const int i = lane;
            double t[32], ls[32];
#pragma clang loop unroll(full)
            for (int c = 0; c < 32; ++c) t[c] = T[taddr(r0 + i, r0 + c)];
            int fail = -1;
            gauss_jordan32(t, ls, i, fail);
            if (fail >= 0 && i == 0) atomicCAS(info, 0, base + r0 + fail + 1);
#pragma clang loop unroll(full)
            for (int c = 0; c < 32; ++c) {
                T[taddr(r0 + i, r0 + c)] = (c <= i) ? t[c] : 0.0;                  // D_j replaces the block
                if (c <= i) A[(int64_t)(r0 + i) * lda + r0 + c] = ls[c];           // L_jj is final
            }
        }
        __syncthreads();
        POTRF_STAMP(2 + 3 * jb);
        const int r1 = r0 + 32;
        const int nr16 = (128 - r1) / 16;
        // rows below: X = A_panel D_j^T, one 16-row strip (two 16x16 tiles) per wave at a time
        // (operands: one base per 16-row strip / 16x16 tile -- a strip or tile never straddles a 32x32 block -- plus constants)
        const double *Dj = T + taddr(r0, r0) + l15 * BS + l4;                  // D_j, B fragment rows l15 / 16 + l15
        // (dealing the 16x16 tiles instead of whole strips to the waves needs a barrier before the in-place write: no gain)
        for (int rt = wave; rt < nr16; rt += 4) {
            const int rb = r1 + 16 * rt;
            double *strip = T + taddr(rb, r0);
            const double *a = strip + l15 * BS + l4;
            d4v x0 = zero4, x1 = zero4;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const double av = a[4 * ks];
                x0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Dj[4 * ks], x0, 0, 0, 0);
                x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, Dj[16 * BS + 4 * ks], x1, 0, 0, 0);
            }
            double *dst = strip + l4 * BS + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dst[4 * r * BS] = x0[r];
                dst[4 * r * BS + 16] = x1[r];
            }
        }
        __syncthreads();
        POTRF_STAMP(3 + 3 * jb);
        // Schur complement: lower 16x16 tiles of T[r1:, r1:] -= X X^T.  Tiles 0..2 are the NEXT diagonal 32x32 block, the only
        // ones on the critical path: the bulk of the update runs beside the next diagonal step (what wave 0 touches there is
        // disjoint from what the others still read and write) and meets it at the barrier after that step.
        const int ntile = nr16 * (nr16 + 1) / 2;
        auto schur_tile = [&](int tt) {
            int ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= tt) ++ti;
            const int tj = tt - ti * (ti + 1) / 2;
            const double *a = T + taddr(r1 + 16 * ti, r0) + l15 * BS + l4, *b = T + taddr(r1 + 16 * tj, r0) + l15 * BS + l4;
            d4v p = zero4;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) p = __builtin_amdgcn_mfma_f64_16x16x4f64(a[4 * ks], b[4 * ks], p, 0, 0, 0);
            double *c = T + taddr(r1 + 16 * ti, r1 + 16 * tj) + l4 * BS + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) c[4 * r * BS] -= p[r];
        };
        auto tile_ij = [](int tt, int &ti, int &tj) {
            ti = 0;
            while ((ti + 1) * (ti + 2) / 2 <= tt) ++ti;
            tj = tt - ti * (ti + 1) / 2;
        };
        auto schur_tile2 = [&](int ta, int tb) {
            int ai, aj, bi, bj;
            tile_ij(ta, ai, aj);
            tile_ij(tb, bi, bj);
            const double *a0 = T + taddr(r1 + 16 * ai, r0) + l15 * BS + l4, *b0 = T + taddr(r1 + 16 * aj, r0) + l15 * BS + l4;
            const double *a1 = T + taddr(r1 + 16 * bi, r0) + l15 * BS + l4, *b1 = T + taddr(r1 + 16 * bj, r0) + l15 * BS + l4;
            d4v p = zero4, q = zero4;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                p = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[4 * ks], b0[4 * ks], p, 0, 0, 0);
                q = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[4 * ks], b1[4 * ks], q, 0, 0, 0);
            }
            double *c0 = T + taddr(r1 + 16 * ai, r1 + 16 * aj) + l4 * BS + l15, *c1 = T + taddr(r1 + 16 * bi, r1 + 16 * bj) + l4 * BS + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                c0[4 * r * BS] -= p[r];
                c1[4 * r * BS] -= q[r];
            }
        };
#ifdef TGP_POTRF_SCHUR_SOLO          // A/B: wave 0 alone on the three tiles of the next diagonal block, no barrier (round 1)
        if (wave == 0) {
            if (nr16 >= 2) {
                const double *x = T + taddr(r1, r0) + l15 * BS + l4;           // X rows r1 + l15 and r1 + 16 + l15 (one block)
                d4v p00 = zero4, p10 = zero4, p11 = zero4;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const double x0 = x[4 * ks];
                    const double x1 = x[16 * BS + 4 * ks];
                    p00 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, p00, 0, 0, 0);
                    p10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x0, p10, 0, 0, 0);
                    p11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, p11, 0, 0, 0);
                }
                double *c = T + taddr(r1, r1) + l4 * BS + l15;                 // the next diagonal block
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    c[4 * r * BS] -= p00[r];
                    c[(16 + 4 * r) * BS] -= p10[r];
                    c[(16 + 4 * r) * BS + 16] -= p11[r];
                }
            }
        } else {
            for (int tt = 3 + (wave - 1); tt < ntile; tt += 3) schur_tile(tt);
        }
#else
        // The three tiles of the next diagonal block go to three waves (one 8-MFMA chain each instead of three on wave 0's SIMD),
        // the fourth wave starts on the rest; after one barrier wave 0 factors that block while waves 1..3 finish the update.
        if (ntile > 0) {
            if (wave < ntile && wave < 4) schur_tile(wave);
            __syncthreads();
            if (wave > 0) {
                // two tiles at a time: two independent 8-MFMA chains hide each other's operand reads and read-modify-write
                int tt = 4 + (wave - 1);
                for (; tt + 3 < ntile; tt += 6) schur_tile2(tt, tt + 3);
                if (tt < ntile) schur_tile(tt);
            }
        }
#endif
        // ---- the inversion's share of this iteration, under the next diagonal step (waves 1..3; see above) ----
        if (DIAG16 && wave > 0 && jb < 3) inversion_step(T, A, lda, info, jb, wave, lane);
    }
    __syncthreads();

    POTRF_STAMP(13);
    if (DIAG16) {
        // what is left of the inversion once D3 exists: one product per wave
        POTRF_STAMP(14);
        {   // waves 1, 2: their slices of W30 and W32; waves 0, 3: the two slices of W31
            const double *D3 = T + blk_origin(3, 3);
            const int h = (wave == 0) ? 0 : (wave == 3 ? 1 : wave - 1);
            d4v acc[2];
            double *blk = T + ((wave == 0 || wave == 3) ? blk_origin(3, 1) : blk_origin(3, 0));
            slice_zero(acc);
            slice_mm<true, false>(D3, blk, h, acc, l15, l4);                                               // W3x = -D3 S3x
            slice_store<true>(blk, h, acc, l15, l4);
            if (wave == 1 || wave == 2) {
                blk = T + blk_origin(3, 2);
                slice_zero(acc);
                slice_mm<true, false>(D3, blk, h, acc, l15, l4);
                slice_store<true>(blk, h, acc, l15, l4);
            }
        }
        __syncthreads();
    } else {
        // off-diagonal blocks of L are final: write them out (diagonal blocks went out from registers).
        // Block row bi holds 32 rows x 32 bi columns = 16 bi column pairs per row; 12 pairs per thread in all.
#pragma unroll
        for (int s = 0; s < 12; ++s) {
            const int bi = s < 2 ? 1 : (s < 6 ? 2 : 3);
            const int first = s < 2 ? 0 : (s < 6 ? 2 : 6);
            const int local = tid + 256 * (s - first);
            const int row = 32 * bi + local / (16 * bi), c = 2 * (local % (16 * bi));
            double2 v;
            v.x = T[taddr(row, c)];
            v.y = T[taddr(row, c + 1)];
            *reinterpret_cast<double2 *>(A + (int64_t)row * lda + c) = v;
        }
        __syncthreads();
        POTRF_STAMP(14);
        // phase 2 of the !DIAG16 reference path: W = L^-1 in place, block column by block column, all four waves per pair
        const int rt = wave >> 1, ct = wave & 1;           // this wave's 16x16 tile of a 32x32 block
        phase2_pair<1, 0>(T, SC, rt, ct, l15, l4);
        phase2_pair<2, 0>(T, SC, rt, ct, l15, l4);
        phase2_pair<3, 0>(T, SC, rt, ct, l15, l4);
        phase2_pair<2, 1>(T, SC, rt, ct, l15, l4);
        phase2_pair<3, 1>(T, SC, rt, ct, l15, l4);
        phase2_pair<3, 2>(T, SC, rt, ct, l15, l4);
    }
    POTRF_STAMP(15);
    {   // W leaves block by block: lower blocks from the LDS image (diagonal ones masked), zeros above
        double *wbase = W + lrow * 128 + 2 * lpair;
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
            for (int bj = 0; bj < 4; ++bj)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double2 x = make_double2(0.0, 0.0);
                    if (bj <= bi) {
                        x = *reinterpret_cast<const double2 *>(tbase + blk_origin(bi, bj) + 16 * h * BS);
                        if (bi == bj) {
                            const int row = lrow + 16 * h, c = 2 * lpair;
                            x.x = (c <= row) ? x.x : 0.0;
                            x.y = (c + 1 <= row) ? x.y : 0.0;
                        }
                    }
                    *reinterpret_cast<double2 *>(wbase + (32 * bi + 16 * h) * 128 + 32 * bj) = x;
                }
    }
    POTRF_STAMP(16);
#ifdef TGP_POTRF_STAMPS
    if (threadIdx.x == 0)      // where it ran: XCC_ID << 8 | HW_ID[15:8]
        tgp_potrf_stamps[((base >> 7) & 1023) * 20 + 17] = (__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 8) | __builtin_amdgcn_s_getreg(4 | (8 << 6) | (7 << 11));
#endif
}

// The LDS image is DYNAMIC shared memory (POTRF_LDS_BYTES, plus a pad where the launch asks for a compute unit of its own) and the
// launch bounds name two workgroups per CU: with a 96 KB static array the compiler knows that one workgroup fits, gives the register
// allocator all 512 registers and it keeps every hoisted LDS address alive (256 architectural VGPRs + 56 accumulators) -- but the
// kernel has to fit on a SIMD beside a wave of the trailing update (264, tools/check_potrf_regs.sh).  Told "two per CU" the allocator
// stays within 256 by rematerialising addresses, without a spill.
constexpr unsigned POTRF_LDS_BYTES = POTRF_LDS_DOUBLES * sizeof(double);
extern __shared__ __attribute__((aligned(16))) double potrf_lds_image[];
template <bool DIAG16>
__global__ __launch_bounds__(256, 2) void potrf128_kernel(double *A, int lda, double *W, int *info, int base) {
    TGP_CHAIN_PRIO();
    potrf128_body<DIAG16>(potrf_lds_image, A, lda, W, info, base);
}
// The same without the register cap (312 registers, no spills: 1 - 2 us faster per block in situ), for launches that get a compute
// unit to themselves anyway: beside a queued bulk update (clear CUs, 128 KB of LDS asked for), or with nothing else on the chip.
__global__ __launch_bounds__(256) void potrf128_solo_kernel(double *A, int lda, double *W, int *info, int base) {
    TGP_CHAIN_PRIO();
    potrf128_body<true>(potrf_lds_image, A, lda, W, info, base);
}
}  // namespace potrf_v2
