// Device-pointer entry points for the multi-GPU (one process per GPU) driver, treegp_amd/dist.py.
// Row-block-cyclic layout over 256-row blocks, reflected every G blocks (dist_owner, tgp_internal.h); a rank's part
// of panel p (blocks b >= p) is stored like the single-GPU panel, back to back, at d_loff[p].
// Nothing here synchronises: every call only enqueues work on the context's stream (which the
// driver points at torch's current stream with tgp_set_stream, so RCCL collectives order with it).
#include "tgp_internal.h"

extern "C" {

int tgp_set_stream(tgp_ctx *ctx, void *stream) {
    if (!ctx) return -1;
    static_cast<void>(hipSetDevice(ctx->device));
    ctx->ext_stream = true;
    ctx->stream = (hipStream_t)stream;          // NULL is the device's default stream
    return 0;
}

// The look-ahead stream of this context's single-GPU schedules (launch_potrf) is the caller's too: the multi-GPU driver lends
// its own idle chain stream to the replicated finish instead of letting the context create one more -- the runtime maps
// streams onto 8 hardware queues, and one stream too many made the driver's bulk and chain streams share a queue
// (rank slice at 8 ranks: chain 75 -> 93 ms).  The context never destroys a stream it was lent; NULL takes it back.
int tgp_set_side_stream(tgp_ctx *ctx, void *stream) {
    if (!ctx) return -1;
    if (!stream) {                                  // take a lent stream back (the context creates its own on next need)
        if (ctx->ext_side_stream) ctx->side_stream = nullptr;
        ctx->ext_side_stream = false;
        return 0;
    }
    if (ctx->side_stream && !ctx->ext_side_stream) (void)hipStreamDestroy(ctx->side_stream);
    ctx->side_stream = (hipStream_t)stream;
    ctx->ext_side_stream = true;
    return 0;
}

int tgp_reset_stream(tgp_ctx *ctx) {
    if (!ctx) return -1;
    ctx->ext_stream = false;
    ctx->stream = ctx->own_stream;
    return 0;
}

int64_t tgp_dist_panel_rows(int64_t p, int64_t Np, int G, int g) {
    return dist_panel_blocks(p, Np / TGP_PW, g, G) * TGP_PW;
}

int64_t tgp_dist_panel_off(int64_t p, int64_t Np, int G, int g) {
    int64_t off = 0;
    const int64_t nB = Np / TGP_PW;
    for (int64_t q = 0; q < p && q < nB; ++q) off += dist_panel_blocks(q, nB, g, G) * TGP_PW * TGP_PW;
    return off;
}

int64_t tgp_dist_local_elems(int64_t Np, int G, int g) { return tgp_dist_panel_off(Np / TGP_PW, Np, G, g); }

int tgp_dd_kbuild(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_yerr,
                  double *d_Aloc, const int64_t *d_loff, int G, int g) {
    TGP_ARG(k && d_X && d_Aloc && d_loff && n > 0);
    return launch_kbuild_lower_dist(ctx, k, d_X, n, padded_n(n), d_yerr, d_Aloc, d_loff, G, g);
}

// owner of panel kpanel: factor its 256x256 diagonal block and pack [L_kk | W0 | W1] (98304 doubles)
int tgp_dd_factor_diag(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, int64_t Np, int kpanel, int G, int g,
                       double *d_W, double *d_bcast) {
    TGP_ARG(d_Aloc && h_loff && d_W && d_bcast && G >= 1 && dist_owner(kpanel, G) == g);
    hipStream_t st = ctx->stream;
    double *blk = d_Aloc + h_loff[kpanel];
    double *W0 = d_W + (int64_t)(2 * kpanel) * TGP_TB * TGP_TB;
    // (latency form where the rank's rows below hold at most 24 row tiles: the chain-bound phase, as for its panel solve)
    const bool latency = dist_panel_blocks(kpanel + 1, Np / TGP_PW, g, G) <= 12;
    int rc = launch_factor_diag256(ctx, blk, W0, W0 + TGP_TB * TGP_TB, kpanel * TGP_PW, latency);
    if (rc) return rc;
    TGP_HIP(hipMemcpyAsync(d_bcast, blk, (size_t)TGP_PW * TGP_PW * 8, hipMemcpyDeviceToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_bcast + TGP_PW * TGP_PW, W0, (size_t)2 * TGP_TB * TGP_TB * 8, hipMemcpyDeviceToDevice, st));
    return 0;
}

// every rank: keep W0/W1 of panel kpanel and solve its own rows below the diagonal block
int tgp_dd_trsm(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, int64_t Np, int kpanel, int G, int g,
                double *d_W, const double *d_bcast) {
    TGP_ARG(d_Aloc && h_loff && d_W && d_bcast);
    double *W0 = d_W + (int64_t)(2 * kpanel) * TGP_TB * TGP_TB;
    double *keepW = dist_owner(kpanel, G) != g ? W0 : nullptr;      // a receiver keeps the inverted blocks: copied by the solve's grid
    const int64_t nB = Np / TGP_PW;
    const int64_t below = dist_panel_blocks(kpanel + 1, nB, g, G);
    const int64_t skip = (dist_owner(kpanel, G) == g) ? TGP_PW : 0;          // the owner's diagonal block comes first
    double *rows = d_Aloc + h_loff[kpanel] + skip * TGP_PW;
    return launch_trsm_rows(ctx, rows, (int)(2 * below), d_bcast, d_bcast + TGP_PW * TGP_PW,
                            d_bcast + TGP_PW * TGP_PW + TGP_TB * TGP_TB, keepW);
}

// tgp_dd_update / tgp_dd_update2: the one- and two-panel forms of tgp_dd_update_group
int tgp_dd_update(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                  const double *d_gathered, int cmax, int col_lo, int col_hi) {
    TGP_ARG(d_Aloc && d_loff && d_gathered && cmax >= 0);
    const double *P[1] = {d_gathered};
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kpanel, G, g, 1, P, &cmax, col_lo, col_hi);
}

int tgp_dd_update2(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                   const double *d_gathered0, int cmax0, const double *d_gathered1, int cmax1, int col_lo, int col_hi) {
    TGP_ARG(d_Aloc && d_loff && d_gathered0 && d_gathered1 && cmax0 >= 0 && cmax1 >= 0);
    const double *P[2] = {d_gathered0, d_gathered1};
    const int cm[2] = {cmax0, cmax1};
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kpanel, G, g, 2, P, cm, col_lo, col_hi);
}

int tgp_dd_update_group(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                        const double *const *d_gathered, const int *cmax, int col_lo, int col_hi) {
    TGP_ARG(d_Aloc && d_loff && d_gathered && cmax && nseg >= 1 && nseg <= 4);
    for (int s = 0; s < nseg; ++s) TGP_ARG(d_gathered[s] && cmax[s] >= 0);
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kpanel, G, g, nseg, d_gathered, cmax, col_lo, col_hi);
}

// Left-looking strip of the panel chain with the panel exchange OFF the chain (dist.py: TGP_DIST_CHAIN_BCAST): block b = kgroup + j
// (1 <= j <= 3) is brought up to date against the j earlier panels of its group, kgroup .. b-1, in one pass of depth 256 j -- this
// rank's rows (blocks >= b) of b's two tile columns.  The ROW operand is the rank's own rows of those panels, read where they are
// stored; the COLUMN operand is block b's rows of them: d_ext[s] (256 x 256, s = panel - kgroup), which the owner of b appends to
// the broadcast of its diagonal block; d_ext == NULL on the owner itself, whose own rows they are.  No all-gathered panel is
// read: the panel chain no longer waits for the exchange of whole panels, which then feeds the bulk update only.
// (Same kernel as tgp_dd_update_group: gathered panel s is [rank][cmax][256][256] and the kernel reads slot g for the rows and
//  slot owner(b) for the columns, so the two bases are placed such that those slots fall on the local rows / on d_ext[s].)
int tgp_dd_strip_left(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, const int64_t *d_loff, int64_t Np, int kgroup, int b, int G,
                      int g, const double *d_ext) {
    const int j = b - kgroup;
    TGP_ARG(d_Aloc && h_loff && d_loff && j >= 1 && j <= 3 && G >= 1 && g >= 0 && g < G && Np % TGP_PW == 0 && b < Np / TGP_PW);
    const int64_t nB = Np / TGP_PW, blk = (int64_t)TGP_PW * TGP_PW;
    const int rj = dist_owner(b, G);
    TGP_ARG(d_ext || rj == g);
    const double *PA[3], *PB[3];
    int cm[3];
    for (int s = 0; s < j; ++s) {
        const int m = kgroup + s;
        int64_t c = 0;                                              // most blocks > m any rank holds (dist.py: panel_cmax)
        for (int r = 0; r < G; ++r) {
            const int64_t n = dist_panel_blocks(m + 1, nB, r, G);
            c = n > c ? n : c;
        }
        cm[s] = (int)c;
        // this rank's rows of panel m below its diagonal block, as it would send them (dist.py: panel_send_view)
        const double *mine = d_Aloc + h_loff[m] + (dist_owner(m, G) == g ? blk : 0);
        PA[s] = mine - (int64_t)g * c * blk;
        const int64_t js = b / G - dist_first_round(m + 1, rj, G);  // block b among owner(b)'s blocks > m
        PB[s] = (rj == g && !d_ext) ? PA[s] : d_ext + s * blk - ((int64_t)rj * c + js) * blk;
    }
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kgroup, G, g, j, PA, cm, 0, 2, 0, 0, PB);
}

// The bulk update as a persistent grid that keeps `nres` (1 .. 3) compute units per shader engine clear for the panel chain
// running on another stream / context (chol.hip: syrk_distn_queue_kernel); tgp_dd_queue_reset once per factorisation, on
// the stream of the context that launches the bulk; tgp_dd_set_exclusive(ctx, 1) on the CHAIN's context makes its diagonal
// blocks insist on a compute unit of their own (safe only while such a bulk keeps units clear, or on an idle chip).
int tgp_dd_queue_reset(tgp_ctx *ctx) {
    TGP_HIP(hipMemsetAsync(ctx->d_queue, 0, TGP_NQUEUE * TGP_QUEUE_WORDS * sizeof(unsigned), ctx->stream));
    ctx->dist_nqueue = 0;
    return 0;
}
int tgp_dd_set_exclusive(tgp_ctx *ctx, int on) {
    if (!ctx) return -1;
    ctx->chain_exclusive = on ? 1 : 0;
    return 0;
}
int tgp_dd_update_group_queued(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                               const double *const *d_gathered, const int *cmax, int col_lo, int col_hi, int nres) {
    TGP_ARG(d_Aloc && d_loff && d_gathered && cmax && nseg >= 1 && nseg <= 4 && nres >= 0);
    for (int s = 0; s < nseg; ++s) TGP_ARG(d_gathered[s] && cmax[s] >= 0);
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kpanel, G, g, nseg, d_gathered, cmax, col_lo, col_hi, nres);
}

// The whole update after a group in ONE launch: the tile columns [0, head_cols) -- the next group's panels, which the panel
// chain waits for -- come first in the workgroup order, the rest follows without a launch boundary (no second ramp and
// tail: at 8 ranks and N = 65 536 a rank's launches hold 2 - 30 rounds of tiles and lost ~1.2 tile times each).  The
// workgroup that finishes the last head tile publishes a sequence number; tgp_dd_wait_head parks another context's stream
// (the chain's) on it.  Needs hand-offs by stream wait-value (tgp_handoff_mode(ctx) == 1); otherwise the caller splits
// the update into two tgp_dd_update_group calls with an event between them.  nres > 0: the persistent-grid form.
int tgp_dd_update_group_fused(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                              const double *const *d_gathered, const int *cmax, int head_cols, int nres) {
    TGP_ARG(d_Aloc && d_loff && d_gathered && cmax && nseg >= 1 && nseg <= 4 && nres >= 0 && head_cols > 0);
    for (int s = 0; s < nseg; ++s) TGP_ARG(d_gathered[s] && cmax[s] >= 0);
    TGP_ARG(tgp_handoff_by_flags(ctx));
    return launch_syrk_distn(ctx, d_Aloc, d_loff, Np, kpanel, G, g, nseg, d_gathered, cmax, 0, -1, nres, head_cols);
}
// the stream of `waiter` proceeds once the head of the last tgp_dd_update_group_fused launched through `ctx` is done
int tgp_dd_wait_head(tgp_ctx *ctx, tgp_ctx *waiter) {
    TGP_ARG(waiter && tgp_handoff_by_flags(ctx));
    TGP_HIP(hipStreamWaitValue32(waiter->stream, ctx->d_flags + 16 * TGP_FLAG_HEAD, ctx->flag_seq[TGP_FLAG_HEAD],
                                 hipStreamWaitValueGte, 0xffffffffu));
    return 0;
}

// ---- replicated factor for the solves ---------------------------------------------------------------------------
// Every rank sees every panel once (diagonal block in the broadcast, the rows below in the all-gather).  Keeping them,
// in the single-GPU packed layout (17 GB at N = 65536, 69 GB at 131072: what 288 GB of HBM per GPU are for), lets each
// rank run the two triangular sweeps locally (tgp_d_potrs) instead of 2 N/256 latency-bound collectives.
namespace {
// one workgroup per 2048-double slice of a 256x256 block: blockIdx.x = block index above kpanel, blockIdx.y = slice
__global__ __launch_bounds__(256) void keep_rows_kernel(const double *__restrict__ gathered, int cmax, int kpanel, int G,
                                                        double *__restrict__ panel) {
    const int64_t b = (int64_t)kpanel + 1 + blockIdx.x;
    const int r = dist_owner(b, G);
    const int64_t idx = b / G - dist_first_round(kpanel + 1, r, G);
    const double2 *src = reinterpret_cast<const double2 *>(gathered + ((int64_t)r * cmax + idx) * TGP_PW * TGP_PW) + blockIdx.y * 1024;
    double2 *dst = reinterpret_cast<double2 *>(panel + (b - kpanel) * TGP_PW * TGP_PW) + blockIdx.y * 1024;
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[threadIdx.x + 256 * u] = src[threadIdx.x + 256 * u];
}
}  // namespace

int tgp_dd_keep_panel(tgp_ctx *ctx, double *d_Afull, int64_t Np, int kpanel, int G, const double *d_bcast,
                      const double *d_gathered, int cmax) {
    TGP_ARG(d_Afull && Np % TGP_PW == 0 && kpanel >= 0 && kpanel < Np / TGP_PW && G >= 1);
    hipStream_t st = ctx->stream;
    double *panel = d_Afull + panel_off(kpanel, Np);
    if (d_bcast) TGP_HIP(hipMemcpyAsync(panel, d_bcast, (size_t)TGP_PW * TGP_PW * 8, hipMemcpyDeviceToDevice, st));
    const int64_t above = Np / TGP_PW - kpanel - 1;
    if (d_gathered && above > 0) {
        keep_rows_kernel<<<dim3((unsigned)above, 32), 256, 0, st>>>(d_gathered, cmax, kpanel, G, panel);
        TGP_HIP(hipGetLastError());
    }
    return 0;
}

// ---- replicated finish -----------------------------------------------------------------------------------------------
// The last `m` rows of the factorisation (m / 256 <= a few dozen blocks) are chain-bound on every rank: per panel a diagonal
// block, a broadcast, local solves and an all-gather, for a bulk update of a handful of tiles.  Instead the ranks exchange
// their shares of the (fully updated) trailing matrix in ONE all-gather, every rank assembles the packed matrix of order m
// and factors it with the single-GPU schedule (launch_potrf), redundantly: same volume over the links, one collective
// instead of 2 m / 256, and 6 ms of arithmetic at m = 8192 where the distributed chain needs ~10.
// `gathered`: [G][stride] doubles, rank r's region = its share from panel k0 on exactly as stored (panels in order, own blocks
// >= p in order).  `dst`: where panel k0 of the packed matrix starts (the trailing part of a packed matrix from panel k0 on is
// itself a packed matrix of order m: the replicated factor's tail, or a buffer of tgp_panel_elems(m)).
}  // extern "C"
namespace {
__device__ __forceinline__ int64_t tail_region_off(int p, int k0, int64_t nB, int r, int G) {      // of panel p inside rank r's region
    int64_t blocks = 0;
    for (int q = k0; q < p; ++q) blocks += dist_panel_blocks(q, nB, r, G);
    return blocks * TGP_PW * TGP_PW;
}
// blockIdx: x = b - k0, y = p - k0, z = 2048-double slice of the block.  TO_SHARE: the opposite direction, own blocks only.
template <bool TO_SHARE>
__global__ __launch_bounds__(256) void tail_blocks_kernel(double *gathered_or_share, int64_t stride, const int64_t *__restrict__ loff,
                                                          int k0, int64_t nB, int G, int g, double *tail) {
    const int b = k0 + blockIdx.x, p = k0 + blockIdx.y;
    if (p > b) return;
    const int r = dist_owner(b, G);
    const int64_t idx = b / G - dist_first_round(p, r, G);
    const int64_t m = (nB - k0) * TGP_PW;
    double2 *packed = reinterpret_cast<double2 *>(tail + panel_off(p - k0, m) + (int64_t)(b - p) * TGP_PW * TGP_PW) + blockIdx.z * 1024;
    if constexpr (TO_SHARE) {
        if (r != g) return;
        double2 *mine = reinterpret_cast<double2 *>(gathered_or_share + loff[p] + idx * TGP_PW * TGP_PW) + blockIdx.z * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) mine[threadIdx.x + 256 * u] = packed[threadIdx.x + 256 * u];
    } else {
        const double2 *src = reinterpret_cast<const double2 *>(gathered_or_share + (int64_t)r * stride + tail_region_off(p, k0, nB, r, G) +
                                                               idx * TGP_PW * TGP_PW) + blockIdx.z * 1024;
#pragma unroll
        for (int u = 0; u < 4; ++u) packed[threadIdx.x + 256 * u] = src[threadIdx.x + 256 * u];
    }
}
}  // namespace
extern "C" {

int tgp_dd_tail_assemble(tgp_ctx *ctx, const double *d_gathered, int64_t stride, int64_t Np, int k0, int G, double *d_tail) {
    const int64_t nB = Np / TGP_PW;
    TGP_ARG(d_gathered && d_tail && G >= 1 && k0 >= 0 && k0 < nB && stride > 0);
    const unsigned nt = (unsigned)(nB - k0);
    tail_blocks_kernel<false><<<dim3(nt, nt, 32), 256, 0, ctx->stream>>>(const_cast<double *>(d_gathered), stride, nullptr, k0, nB, G, 0, d_tail);
    TGP_HIP(hipGetLastError());
    return 0;
}
// the factored tail back into this rank's share (its sweeps and its log-determinant read the share)
int tgp_dd_tail_scatter(tgp_ctx *ctx, const double *d_tail, int64_t Np, int k0, int G, int g, double *d_Aloc, const int64_t *d_loff) {
    const int64_t nB = Np / TGP_PW;
    TGP_ARG(d_tail && d_Aloc && d_loff && G >= 1 && g >= 0 && g < G && k0 >= 0 && k0 < nB);
    const unsigned nt = (unsigned)(nB - k0);
    tail_blocks_kernel<true><<<dim3(nt, nt, 32), 256, 0, ctx->stream>>>(d_Aloc, 0, d_loff, k0, nB, G, g, const_cast<double *>(d_tail));
    TGP_HIP(hipGetLastError());
    return 0;
}

// forward sweep, block kb (owner): y_k (256) <- L_kk^-1 y_k
int tgp_dd_fwd_diag(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int kb, const double *d_W, double *d_yk) {
    const double *W0 = d_W + (int64_t)(2 * kb) * TGP_TB * TGP_TB;
    return launch_diag256_fwd(ctx, d_Aloc + h_loff[kb], W0, W0 + TGP_TB * TGP_TB, d_yk);
}
// forward sweep, every rank: local rows of blocks > kb:  y_loc -= L[:, kb] z_k
int tgp_dd_fwd_update(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int64_t Np, int kb, int G, int g,
                      const double *d_zk, double *d_yloc) {
    const int64_t nB = Np / TGP_PW;
    const int64_t below = dist_panel_blocks(kb + 1, nB, g, G);
    if (below <= 0) return 0;
    const int64_t skip = (dist_owner(kb, G) == g) ? TGP_PW : 0;
    const int64_t lb0 = dist_first_round(kb + 1, g, G);                 // local index of the first block > kb
    return launch_fwd_update_rows(ctx, d_Aloc + h_loff[kb] + skip * TGP_PW, below * TGP_PW, d_zk, d_yloc + lb0 * TGP_PW);
}
// backward sweep, every rank: s (256) = sum over local rows of blocks > kb of L[i, kb]^T a_i
int tgp_dd_bwd_partial(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int64_t Np, int kb, int G, int g,
                       const double *d_aloc, double *d_s) {
    const int64_t nB = Np / TGP_PW;
    const int64_t below = dist_panel_blocks(kb + 1, nB, g, G);
    const int64_t skip = (dist_owner(kb, G) == g) ? TGP_PW : 0;
    const int64_t lb0 = (below > 0) ? dist_first_round(kb + 1, g, G) : 0;
    return launch_gemv_t_rows(ctx, d_Aloc + h_loff[kb] + skip * TGP_PW, below * TGP_PW, d_aloc + lb0 * TGP_PW, d_s);
}
// backward sweep, block kb (owner): a_k (256) <- L_kk^-T (a_k - s)   (d_s may be NULL)
int tgp_dd_bwd_diag(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int kb, const double *d_W, double *d_ak,
                    const double *d_s) {
    const double *W0 = d_W + (int64_t)(2 * kb) * TGP_TB * TGP_TB;
    return launch_diag256_bwd(ctx, d_Aloc + h_loff[kb], W0, W0 + TGP_TB * TGP_TB, d_ak, d_s);
}

int tgp_dd_logdet_local(tgp_ctx *ctx, const double *d_Aloc, const int64_t *d_loff, int64_t Np, int64_t n, int G, int g,
                        double *d_out) {
    return launch_logdet_dist(ctx, d_Aloc, d_loff, Np, n, G, g, d_out);
}

// first non-positive pivot seen since the last reset (0 = none); synchronises the stream
int tgp_dd_info(tgp_ctx *ctx, int reset) {
    hipStream_t st = ctx->stream;
    TGP_HIP(hipMemcpyAsync(ctx->h_info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    const int info = *ctx->h_info;
    if (reset) TGP_HIP(hipMemsetAsync(ctx->d_info, 0, sizeof(int), st));
    return info;
}

}  // extern "C"
