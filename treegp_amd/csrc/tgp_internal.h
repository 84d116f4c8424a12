// Internal declarations shared by the HIP translation units of libtgp.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <atomic>
#include <vector>
#include "../../include/tgp.h"

// Kernels of the serial panel chain (diagonal block, panel solves, strips) share compute units with the bulk trailing
// update during the look-ahead: their waves ask the instruction arbiter for precedence over the bulk's.
#ifdef TGP_NO_CHAIN_PRIO
#define TGP_CHAIN_PRIO() ((void)0)
#else
#define TGP_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)
#endif

#define TGP_QUEUE_MAXRES 6    // most compute units per shader engine and XCD (of 8) a queued bulk update can keep clear
#define TGP_QUEUE_LEAVE (8 + 32 * TGP_QUEUE_MAXRES)      // word offset of the 8 leave counters
#define TGP_QUEUE_WORDS (TGP_QUEUE_LEAVE + 8)   // per launch: 8 XCD-class tile counters + 8 x 4 x MAXRES clear-CU words + 8 leave counters
#define TGP_NQUEUE 128        // persistent bulk-update launches per factorisation (one set of 8 counters each)
#define TGP_PSYNC_PANELS 1024 // panels with a counter set of their own (Np <= 262144); 16 words = 64 B each
#define TGP_TB 128            // tile / diagonal-block size
#define TGP_PW 256            // panel width = trailing-update depth

struct tgp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // the stream every kernel is launched on
    hipStream_t own_stream = nullptr;  // created by tgp_init
    hipStream_t side_stream = nullptr; // high-priority stream for the Cholesky look-ahead
    unsigned *d_flags = nullptr;       // cross-stream hand-off flags (handoff.hip: hand-offs by stream wait-value), 16 x 64 B
    unsigned flag_seq[16] = {0};       // last value signalled on each (monotonic over the context's life)
    unsigned head_count = 0;           // value of the head-tile counter (flag word TGP_FLAG_HEAD_COUNT) after the last fused launch
    int handoff = 0;                   // 0 undecided, 1 flags + stream wait-value, 2 events (tgp_handoff_by_flags)
    bool ext_stream = false;           // stream was set by tgp_set_stream
    bool ext_side_stream = false;      // side_stream was lent by tgp_set_side_stream
    std::string err;
    double timings[TGP_NTIMINGS] = {0};
    int profiling = 0;
    int lookahead = 1;                 // 0: factorise on the one stream (tgp_set_lookahead; for contexts that run side by side)
    hipEvent_t ev[8] = {nullptr};
    // grow-only scratch
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    void *scratch2 = nullptr;     // second, small scratch (partial sums) that may be live beside `scratch`
    size_t scratch2_bytes = 0;
    hipEvent_t ev_slab[17] = {nullptr};   // slab-build pipeline (trsv.hip), created on first use
    void *vslab = nullptr;        // inverse slabs of the 1024-row triangular sweeps (trsv_big.hip) when the caller keeps none
    size_t vslab_bytes = 0;
    void *vslab_tt = nullptr;     // build-time scratch of the slab build (Np x S doubles): one per context, not one per factor
    size_t vslab_tt_bytes = 0;
    int dist_nqueue = 0;          // queue sets handed out since tgp_dd_queue_reset (multi-GPU driver)
    int chain_exclusive = 0;      // tgp_dd_set_exclusive: diagonal blocks of this context ask for a compute unit of their own
    unsigned *d_queue = nullptr;  // tile-queue counters of the persistent bulk update (TGP_NQUEUE launches x TGP_QUEUE_WORDS)
    unsigned *d_psync = nullptr;  // in-kernel hand-off counters of panel_mid_kernel (chol.hip): TGP_PSYNC_PANELS x 16 words, zeroed per factorisation
    int mid_off = 0;              // an in-kernel hand-off of this context timed out once: panel_mid_kernel stays off for its life
    int mid_allowed = 0;          // set by the solve entry points for the factorisation they are about to queue (alone on the chip, retry possible)
    int *d_info = nullptr;        // first failing pivot (1-based), 0 = ok
    int *h_info = nullptr;        // pinned mirror
    double *d_scal = nullptr;     // small device scalars (logdet, dot, ...)
    double *h_scal = nullptr;     // pinned mirror (16 doubles)
    std::vector<hipEvent_t> prof_events;
};

struct tgp_factor {
    int64_t n = 0, Np = 0;
    double *d_A = nullptr;        // packed lower panels
    double *d_W = nullptr;        // inverted 128x128 diagonal blocks
    double *d_slabs = nullptr;    // inverse slabs of the big-step sweeps, built by the first solve with this factor
    int slab_S = 0;               // the step they were built for
    double *d_slabs2 = nullptr;   // slabs of another step for the block substitution of cov.hip (factor_slabs)
    int slab2_S = 0;
    bool borrowed = false;        // tgp_factor_borrow: d_A / d_W belong to the caller (multi-GPU driver's replicated factor)
};

#define TGP_HIP(call)                                                                   \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return -2;                                                                  \
        }                                                                               \
    } while (0)

#define TGP_ARG(cond)                                                                   \
    do {                                                                                \
        if (!(cond)) {                                                                  \
            ctx->err = std::string("bad argument: ") + #cond;                           \
            return -1;                                                                  \
        }                                                                               \
    } while (0)

__host__ __device__ inline int64_t panel_off(int64_t p, int64_t Np) {
    return (int64_t)TGP_PW * (p * Np - (int64_t)(TGP_PW / 2) * p * (p - 1));
}
__host__ __device__ inline int64_t padded_n(int64_t n) {
    return (n + TGP_PW - 1) / TGP_PW * TGP_PW;
}

// ---- multi-GPU row-block-cyclic geometry (blocks of 256 rows) ----
// The owner map is the block-cyclic deal REFLECTED every G blocks ("snake"): round q = b / G hands one block to every rank,
// rank r's being q G + r in even rounds and q G + (G-1-r) in odd ones.  Block row b of the lower triangle carries b + 1 block
// columns, so the plain deal b -> b % G gives rank G-1 the heaviest block of every round (4.1 % above the mean at 256 blocks
// on 8 ranks, and the slowest rank sets the pace); the reflection pairs a heavy round with a light one: 0.2 %.
// A rank's local index of block b is b / G either way.  G == 1: the identity.
__host__ __device__ inline int dist_owner(int64_t b, int G) {
    const int64_t q = b / G;
    const int p = (int)(b - q * G);
    return (q & 1) ? G - 1 - p : p;
}
// rank r's block of round q
__host__ __device__ inline int64_t dist_block_of(int64_t q, int r, int G) { return q * G + ((q & 1) ? G - 1 - r : r); }
// round of the smallest block >= s owned by rank r (= its local index)
__host__ __device__ inline int64_t dist_first_round(int64_t s, int r, int G) {
    const int64_t q = s / G;
    return dist_block_of(q, r, G) >= s ? q : q + 1;
}
// smallest block index >= s owned by rank r
__host__ __device__ inline int64_t dist_first_ge(int64_t s, int r, int G) { return dist_block_of(dist_first_round(s, r, G), r, G); }
// number of local 256-row blocks of rank g in panel p (blocks p <= b < nB)
__host__ __device__ inline int64_t dist_panel_blocks(int64_t p, int64_t nB, int g, int G) {
    if (nB <= 0) return 0;
    const int64_t q0 = dist_first_round(p, g, G);
    int64_t ql = (nB - 1) / G;
    if (dist_block_of(ql, g, G) >= nB) --ql;
    return ql >= q0 ? ql - q0 + 1 : 0;
}

// trailing-update tile enumeration (XCD-aware): see chol.hip.  Blocks b, b+8, ... share an XCD; an XCD works through
// whole super-tiles of sup x sup tiles (shared operand rows stay in its L2).  Super-tiles are dealt to the XCDs in turn; whole
// ones only while all eight get one: the last ns % 8 of them would leave the XCDs a full super-tile apart (0.3 % of a launch at
// T = 496 but 7 % at T = 56 and 25 % at T = 40 with 8 x 8; the queued bulk update of the chain-bound sizes has one tile queue
// per XCD class and ends with its slowest), so each of those is cut into eight equal slices, one per XCD (round 4).  Below
// TGP_SUP4_BELOW tile rows the super-tiles are 4 x 4, below 32 tile rows 2 x 2.
#ifndef TGP_SUP4_BELOW
#define TGP_SUP4_BELOW 192
#endif
__host__ __device__ inline int tilemap_sup_shift(int64_t T) { return T < 32 ? 1 : (T < TGP_SUP4_BELOW ? 2 : 3); }
// slots per XCD class: per (= sup^2) for every whole round of eight super-tiles, q = max(per / 8, 1) for each of the rest
__host__ __device__ inline int64_t tilemap_slots(int64_t T) {
    if (T <= 0) return 0;
    const int sh = tilemap_sup_shift(T);
    const int64_t S = (T + (1 << sh) - 1) >> sh;   // super-tiles per side
    const int64_t ns = S * (S + 1) / 2;            // lower-triangular super-tiles
    const int64_t per = (int64_t)1 << (2 * sh), q = per >= 8 ? per / 8 : 1;
    return (ns / 8) * per + (ns % 8) * q;
}
__host__ __device__ inline int64_t tilemap_grid(int64_t T) { return 8 * tilemap_slots(T); }
// block id -> (ti, tj), or ti = -1 when the slot is empty
__host__ __device__ inline void tilemap(int64_t b, int64_t T, int &ti, int &tj) {
    ti = -1; tj = -1;
    if (T <= 0 || b < 0) return;
    const int sh = tilemap_sup_shift(T);
    const int64_t S = (T + (1 << sh) - 1) >> sh;
    const int64_t ns = S * (S + 1) / 2;
    const int per = 1 << (2 * sh), q = per >= 8 ? per / 8 : 1;
    const int xcd = (int)(b & 7);                  // blocks b, b+8, ... share an XCD (speed only)
    const int64_t slot = b >> 3;
    const int64_t whole = (ns / 8) * per;
    int64_t st;
    int within;
    if (slot < whole) {
        st = (slot >> (2 * sh)) * 8 + xcd;         // super-tile handled by this XCD group
        within = (int)(slot & (per - 1));
    } else {                                       // the last ns % 8 super-tiles: slice xcd of each
        const int64_t nn = slot - whole;
        st = (ns / 8) * 8 + nn / q;
        within = xcd * q + (int)(nn % q);
        if (st >= ns || within >= per) return;
    }
    // st -> (Si, Sj), Sj <= Si, row-major triangular enumeration
    int64_t Si = (int64_t)((sqrt(8.0 * (double)st + 1.0) - 1.0) * 0.5);
    while (Si * (Si + 1) / 2 > st) --Si;
    while ((Si + 1) * (Si + 2) / 2 <= st) ++Si;
    const int64_t Sj = st - Si * (Si + 1) / 2;
    const int i = (int)((Si << sh) + (within >> sh));
    const int j = (int)((Sj << sh) + (within & ((1 << sh) - 1)));
    if (j > i || i >= T) return;
    ti = i; tj = j;
}

// implemented across the .hip files
int tgp_ensure_side_stream(tgp_ctx *ctx);
int tgp_ensure_scratch(tgp_ctx *ctx, size_t bytes);
int tgp_ensure_scratch2(tgp_ctx *ctx, size_t bytes);
// grow-only PINNED host scratch of the context (api.hip): host-built tables that are uploaded whole are built in it, so that the
// runtime neither pins the pages of a std::vector for the transfer nor unpins them behind the call
int tgp_ensure_pinned(tgp_ctx *ctx, size_t bytes, void **out);
int launch_kbuild_lower(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, int64_t Np,
                        const double *d_yerr, double *d_A);
int launch_kernel_dense(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n,
                        const double *d_Y, int64_t m, int self, double *d_out);
int launch_potrf(tgp_ctx *ctx, double *d_A, int64_t Np, double *d_W, bool defer_info = false, int64_t n_data = -1);
int tgp_potrf_info_rc(tgp_ctx *ctx, int info);      // device `info` word -> return code (TGP_RC_HANDOFF: in-kernel hand-off timed out)
#define TGP_RC_HANDOFF (-4)
// Solves in flight in this process (host-boundary and device-resident solve calls, counted from entry to return).  Kernels whose
// workgroups wait for each other inside one launch (panel_mid_kernel) are used only by a solve that found itself ALONE when it
// started: workgroups are dealt to the eight XCDs in turn and each XCD starts its share when it has room, so "producers have
// lower indices" guarantees progress for one such kernel on the chip, not for several that hold compute units while they wait
// for workgroups another one keeps out (five contexts side by side did deadlock: bounded wait -> TGP_RC_HANDOFF).  Whoever enters
// while another solve is in flight runs without it, so at most one solve at a time has waiting workgroups.  (Other PROCESSES on
// the same GPU are invisible here: that case is what the bounded wait and the retry in api.hip are for.)
extern std::atomic<int> tgp_solves_in_flight;
// cross-stream hand-offs (handoff.hip).  Flag ids: 0, 1 the look-ahead of launch_potrf; TGP_FLAG_HEAD the "head columns
// done" signal of a fused multi-GPU bulk launch, TGP_FLAG_HEAD_COUNT the word its workgroups count themselves in on.
#define TGP_FLAG_HEAD 8
#define TGP_FLAG_HEAD_COUNT 9
bool tgp_handoff_by_flags(tgp_ctx *ctx);
unsigned tgp_next_seq(tgp_ctx *ctx, int id, hipError_t *err);
hipError_t tgp_signal_value(tgp_ctx *ctx, hipStream_t from, int id, unsigned v);
hipError_t tgp_signal(tgp_ctx *ctx, hipStream_t from, int id, hipEvent_t ev);
hipError_t tgp_await(tgp_ctx *ctx, hipStream_t to, int id, hipEvent_t ev);
bool potrs_big_step(int64_t Np, int *S);
int acquire_slabs(tgp_ctx *ctx, int64_t Np, int S, double **slab_cache, int *slab_S, double **out, bool *need_build);
int launch_potrs_big_bwd(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_b, double *d_z);
int launch_kbuild_lower_dist(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, int64_t Np,
                             const double *d_yerr, double *d_Aloc, const int64_t *d_loff, int G, int g);
int launch_factor_diag256(tgp_ctx *ctx, double *blk, double *W0, double *W1, int base, bool latency = false);
int launch_trsm_rows(tgp_ctx *ctx, double *rows, int ntiles, const double *Lkk, const double *W0, const double *W1, double *keepW = nullptr);
int launch_syrk_distn(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                      const double *const *d_P, const int *cmax, int col_lo, int col_hi, int queue_nres = 0, int head_cols = 0,
                      const double *const *d_PB = nullptr);
int launch_diag256_fwd(tgp_ctx *ctx, const double *Lkk, const double *W0, const double *W1, double *y);
int launch_diag256_bwd(tgp_ctx *ctx, const double *Lkk, const double *W0, const double *W1, double *y, const double *s);
int launch_fwd_update_rows(tgp_ctx *ctx, const double *Lrows, int64_t rows, const double *z, double *yrows);
int launch_gemv_t_rows(tgp_ctx *ctx, const double *Lrows, int64_t rows, const double *a, double *s);
int launch_logdet_dist(tgp_ctx *ctx, const double *d_Aloc, const int64_t *d_loff, int64_t Np, int64_t n, int G, int g,
                       double *d_out);
// slab_cache: where the caller keeps this factor's inverse slabs (built on first use, owned by the caller: hipFree);
// nullptr = rebuild them into the context's own buffer on every call
int launch_potrs(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b, bool forward_only = false,
                 double **slab_cache = nullptr, int *slab_S = nullptr);
int launch_potrs_128(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b, bool forward_only);
size_t vslab_bytes(int64_t Np, int S);
int factor_slabs(tgp_ctx *ctx, tgp_factor *f, int want_S, int *S, const double **slabs);
int tgp_ensure_io(tgp_ctx *ctx, size_t bytes);
void *tgp_io_buffer(tgp_ctx *ctx);
void tgp_factor_release_to_cache(tgp_ctx *ctx, tgp_factor *f);
int launch_loglik_grad(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *d_X, const double *d_alpha, double *grad);
int launch_vslab_build(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, int S, double *slabs);
// slabs being built on the side stream in chunks of `chunk` super-blocks while the forward sweep already runs: ready[c] is
// recorded behind chunk c (chunk 0 is built on the sweep's own stream)
struct SlabPipeline {
    int chunk;
    hipEvent_t ready[16];
};
int launch_vslab_build_range(tgp_ctx *ctx, hipStream_t st, const double *d_A, const double *d_W, int64_t Np, int S, double *slabs,
                             int64_t row_lo, int64_t row_hi);
int launch_potrs_big(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_b, double *d_z,
                     bool forward_only, const SlabPipeline *pipe = nullptr);
int launch_potrs_big_multi(tgp_ctx *ctx, const double *d_A, int64_t Np, int S, const double *slabs, double *d_B, double *d_Z, int nrhs);
int launch_logdet(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out);
int launch_dot(tgp_ctx *ctx, const double *d_a, const double *d_b, int64_t n, double *d_out);
int launch_logdet_dot(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, const double *d_a, const double *d_b, double *d_out);
int launch_pad_copy(tgp_ctx *ctx, const double *d_y, int64_t n, int64_t Np, double *d_b);
int launch_augment_rhs(tgp_ctx *ctx, double *d_A, int64_t Np, int64_t n, const double *d_y);
int launch_extract_row(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_z);
int launch_logdet_rowsq(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out);
int launch_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_alpha,
                   const double *d_Xs, int64_t m, double *d_ys);
int launch_unpack_lower(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *d_out);
int launch_pack_lower(tgp_ctx *ctx, const double *d_K, int64_t n, int64_t Np, const double *d_yerr, double *d_A);
// d_B: (nrhs, Np) right-hand sides, solved in place, the factor read once per sweep for groups of up to 8 of them
int launch_potrs_multi(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_B, int nrhs, double **slab_cache,
                       int *slab_S = nullptr);
