/* tgp.h -- C-ABI of libtgp.so: the MI355X-native GP hot path behind treegp's Python API.
 *
 * The reference (PFLeget/treegp, pure Python) has no FFI of its own; it calls SciPy /
 * scikit-learn / TreeCorr at a handful of seams.  Each entry point below replaces one of those
 * seams and cites the reference lines it stands in for (paths relative to the reference root).
 *
 *   S1  kernel.__call__(X[,Y])           treegp/kernels.py:114-126, 249-276, 355-381
 *   S2  cholesky + cho_solve (+ logdet)  treegp/gp_interp.py:180-182, treegp/log_likelihood.py:29-33
 *   S3  HT @ alpha                       treegp/gp_interp.py:177,183
 *   S3b posterior covariance             treegp/gp_interp.py:184-192
 *   S4  treecorr KKCorrelation.process   treegp/two_pcf.py:297-305, 330-334, 342-362
 *   S5  KNeighborsRegressor.predict      treegp/gp_interp.py:236-238
 *   S6  binned_statistic_2d              treegp/meanify.py:76-107
 *   S7  vcorr pair accumulation          treegp/utils.py:36-72
 * followed by a device-resident tier (bench, tests) and the multi-GPU tier driven by treegp_amd/dist.py.
 *
 * Conventions
 *   - plain C types only; every array is C-contiguous float64 (or int64 where named so).
 *   - pointers are HOST pointers unless the parameter name starts with d_ (device pointer).
 *     The library never keeps a host pointer after the call returns.
 *   - coordinates are always passed as (n, 2) row-major; 1-D problems pass a zero second
 *     column (what treegp/two_pcf.py:250-251 does for the pair counter).
 *   - return 0 = ok; > 0 = LAPACK-style "leading minor of order i is not positive definite"
 *     (the Python shim raises numpy.linalg.LinAlgError, as scipy.linalg.cholesky does at
 *     gp_interp.py:181); < 0 = argument / HIP error, text in tgp_last_error().
 *   - a tgp_ctx owns one device, one stream and a grow-only workspace; it is not re-entrant.
 */
#ifndef TGP_H
#define TGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tgp_ctx tgp_ctx;
typedef struct tgp_factor tgp_factor;

/* kernel kinds (treegp/kernels.py + sklearn RBF); amp = sigma^2 of sklearn's
 * Product(ConstantKernel(sigma^2), k) folded in: value = amp * k(x, x').               */
enum {
    TGP_RBF = 0,      /* sklearn RBF(l): a = c = 1/l^2, b = 0 filled by the caller       */
    TGP_ARBF = 1,     /* AnisotropicRBF: exp(-0.5 q), q = a dx^2 + 2 b dx dy + c dy^2   */
    TGP_VK = 2,       /* VonKarman(l):  u = |d|/l                                       */
    TGP_AVK = 3       /* AnisotropicVonKarman: u = sqrt(q)                              */
};                    /* von Karman: u^(5/6) K_{5/6}(2 pi u) / lim0, 1 at u == 0        */

typedef struct {
    int32_t kind;
    int32_t _pad;
    double amp;       /* sigma^2                                                       */
    double a, b, c;   /* invLam[0,0], invLam[0,1], invLam[1,1]                          */
    double ell;       /* VonKarman length_scale                                         */
} tgp_kernel;

/* ---- context ------------------------------------------------------------------------ */
/* One context = one GPU: ndev must be 1.  SURVEY 8(b)'s "one process drives the node's GPUs" (ndev > 1) is served one level up:
 * treegp_amd/dist_pool.py starts one worker process per GPU, each with a context of its own and an RCCL group among them, behind
 * GPInterpolation(backend="dist") in an ordinary single-process script; the tgp_dd_* entry points below are the
 * per-rank building blocks those workers (and torchrun ranks) call.                                                          */
int tgp_init(const int *devices, int ndev, tgp_ctx **out);
void tgp_destroy(tgp_ctx *ctx);
const char *tgp_last_error(tgp_ctx *ctx);
const char *tgp_version(void);
int tgp_device_count(void);

/* phase timings (ms, hipEvent on the ctx stream) of the last call that recorded them:
 * [0] K build  [1] Cholesky total  [2] triangular solves  [3] predict  [4] pair binning
 * [5] trailing-update (syrk) kernel time summed  [6] number of trailing-update launches
 * [7] trailing-update flops (sum over launches)  [8] K-build bytes written
 * [9] result transfer of tgp_gp_predict_cov ([3] is then its device compute time)
 * [10] triangular sweeps over L inside [2]: 2 = forward + backward, 1 = backward only (the forward substitution
 *      rode along with the factorisation, inside [1]), 0 = none (likelihood only, right-hand side as a matrix row) */
#define TGP_NTIMINGS 11
int tgp_last_timings(tgp_ctx *ctx, double *ms, int n);
/* when on (default off) the Cholesky brackets every trailing-update launch with events */
int tgp_set_profiling(tgp_ctx *ctx, int on);
/* 0: the factorisation of this context stays on its one stream (no look-ahead on the side stream).  For contexts
 * that run side by side -- e.g. the independent likelihood evaluations of one finite-difference gradient
 * (treegp/log_likelihood.py:57 lets SciPy take them one after the other): hardware queues are few.          */
int tgp_set_lookahead(tgp_ctx *ctx, int on);

/* ---- S1: kernel matrix (host buffers) --------------------------------------------------
 * out (n, m) row-major = amp * k(X_i, Y_j).  Y == NULL: self kernel, out is (n, n) and the
 * diagonal is exactly amp (kernels.py:121,261,367).                                      */
int tgp_kernel_matrix(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n,
                      const double *Y, int64_t m, double *out);

/* ---- S2: alpha = (K + diag(yerr^2))^-1 y, logdet = sum 2 log diag(chol) ------------------
 * K never leaves the device.  keep != NULL receives a handle on the device-resident factor
 * (free with tgp_factor_free).  alpha may be NULL (log-likelihood only needs y.alpha, which
 * is returned in *ydota when ydota != NULL; it is then computed as |L^-1 y|^2, with y carried
 * through the factorisation as an extra row of the matrix, or from the forward sweep alone).  */
int tgp_gp_solve(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n,
                 const double *y, const double *yerr, double *alpha, double *logdet,
                 double *ydota, tgp_factor **keep);
void tgp_factor_free(tgp_ctx *ctx, tgp_factor *f);
/* gives the handle up but leaves its device memory with the context as the factor cache of the next solve of the same size
 * (no hipFree + hipMalloc of the packed matrix between two fits of one size); freed by tgp_destroy or a solve of another size */
void tgp_factor_release(tgp_ctx *ctx, tgp_factor *f);
/* the context gives back the device memory it holds between calls (factor cache, inverse slabs, scratch); they are allocated
 * again on demand.  For a process that is about to allocate elsewhere on the same GPU.                                     */
int tgp_release_caches(tgp_ctx *ctx);
/* a handle on a factor in the CALLER's device memory (d_A: tgp_panel_elems(Np) packed doubles, d_W: Np x 128, as
 * tgp_d_potrf leaves them) -- the multi-GPU driver's replicated factor; tgp_factor_free never frees d_A / d_W of it.
 * Serves the same reference lines as a kept factor: gp_interp.py:184-192 (covariance), README.rst:28 (several fields). */
int tgp_factor_borrow(tgp_ctx *ctx, double *d_A, double *d_W, int64_t n, tgp_factor **out);

/* ---- S2b: the same for a kernel matrix the CALLER evaluated ---------------------------------
 * For scikit-learn kernel trees tgp_kernel cannot describe (Sum, WhiteKernel, Matern, ...): the
 * reference evals any of them (treegp/kernels.py:17-59) and factorises what kernel(X1) returns
 * (treegp/gp_interp.py:180-182).  K: (n, n) row-major, its lower triangle is read; yerr^2 (may be
 * NULL) is added to the diagonal on the device.  Everything else as tgp_gp_solve.              */
int tgp_gp_solve_dense(tgp_ctx *ctx, const double *K, int64_t n, const double *y, const double *yerr,
                       double *alpha, double *logdet, double *ydota, tgp_factor **keep);

/* ---- S2c: more right-hand sides against a kept factor ("multi-field") -----------------------
 * One GP per field, all fields sharing the coordinates and the kernel (the Piff pattern the
 * reference names at README.rst:28; with the reference each field is its own GPInterpolation and
 * its own cholesky + cho_solve, gp_interp.py:180-182).  B and Xout: (nrhs, n) row-major;
 * Xout[v] = (K + diag(yerr^2))^-1 B[v].  The factor is read once per sweep for up to 4 fields.  */
int tgp_factor_solve(tgp_ctx *ctx, tgp_factor *f, const double *B, int nrhs, double *Xout);

/* ---- S3: ys[j] = sum_i amp k(Xs_j, X_i) alpha_i, HT never materialised -------------------*/
int tgp_gp_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n,
                   const double *alpha, const double *Xs, int64_t m, double *ys);

/* ---- S3b: cov (m, m) = k(Xs,Xs) - HT K^-1 HT^T using a kept factor ------------------------*/
int tgp_gp_predict_cov(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *X,
                       int64_t n, const double *Xs, int64_t m, double *cov);
/* the same with HT = kernel(X2, Y=X1) (m, n) and Kss = kernel(X2) (m, m) evaluated by the caller
 * (treegp/gp_interp.py:177,191 for kernel trees that go through tgp_gp_solve_dense)              */
int tgp_gp_predict_cov_dense(tgp_ctx *ctx, tgp_factor *f, const double *HT, const double *Kss,
                             int64_t m, double *cov);

/* ---- S2d: gradient of the log marginal likelihood from a kept factor and its alpha ---------
 * (the reference's optimiser passes no jac, treegp/log_likelihood.py:57 -- this is what a caller who wants one
 * gets, in the kernel-derivative convention of treegp/kernels.py:128-150.)
 *   grad[0..3] = 1/2 sum_ij (alpha_i alpha_j - [K^-1]_ij) dK_ij/dp   for p = log amp, a, b, c   (b = invLam[0,1] = invLam[1,0])
 * Gaussian kernels only (TGP_RBF, TGP_ARBF): -1 with a message for the von Karman kinds.  K^-1 is formed on the device
 * (2/3 n^3 flops, two n x n buffers) and never leaves it.                                                             */
int tgp_gp_loglik_grad(tgp_ctx *ctx, tgp_factor *f, const tgp_kernel *k, const double *X, int64_t n,
                       const double *alpha, double *grad);
/* K build + factorisation + solve + the gradient above in one call, for X (n, 2), y, yerr (may be NULL) that live on the
 * device (tgp_dev_alloc / tgp_h2d): what one evaluation of a gradient-driven fit needs; the factor is not kept.        */
int tgp_d_gp_solve_grad(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_y,
                        const double *d_yerr, double *logdet, double *ydota, double *grad);

/* ---- S4: binned scalar pair correlation, exact binning -----------------------------------
 * w == NULL: unit weights.  TwoD: nbins x nbins pixels over [-max_sep, max_sep]^2, outputs
 * of length nbins^2 (flat index iy*nbins+ix).  Log: nbins log-spaced bins in
 * [min_sep, max_sep).  xi = sum(w_i w_j k_i k_j)/sum(w_i w_j), 0 where the weight is 0.     */
int tgp_kk_twod(tgp_ctx *ctx, const double *x, const double *y, const double *k,
                const double *w, int64_t n, double min_sep, double max_sep, int nbins,
                double *xi, double *weight, double *npairs);
int tgp_kk_log(tgp_ctx *ctx, const double *x, const double *y, const double *k,
               const double *w, int64_t n, double min_sep, double max_sep, int nbins,
               double *xi, double *weight, double *meanr, double *meanlogr, double *npairs);
/* One shard of the same pair loop (multi-GPU, SURVEY 8e "pair histogram"): bin_type 0 = TwoD,
 * 1 = Log; the 256-point i-tiles are dealt round-robin to `nparts` callers and `part` bins
 * only its own.  acc_out holds the raw sums, 3 x nbins^2 (sum w w k k, sum w w, pairs) for
 * TwoD and 5 x nbins (sum wwkk, sum ww, sum ww r, sum ww ln r, pairs) for Log; adding the
 * shards (one all-reduce) and dividing by the weights gives tgp_kk_twod / tgp_kk_log.      */
int tgp_kk_partial(tgp_ctx *ctx, int bin_type, const double *x, const double *y, const double *k,
                   const double *w, int64_t n, double min_sep, double max_sep, int nbins,
                   int part, int nparts, double *acc_out);
/* bootstrap (two_pcf.py:269-281, 342-362): idx is (n_boot, n) int64; resample b uses points
 * idx[b,:], k = yv[idx] - mean(yv[idx]), w = 1/yerr[idx]^2 (unit weights if yerr == NULL or
 * sum(yerr[idx]) == 0).  xi_out is (n_boot, nbins^2).                                       */
int tgp_kk_twod_bootstrap(tgp_ctx *ctx, const double *x, const double *y, const double *yv,
                          const double *yerr, int64_t n, const int64_t *idx, int64_t n_boot,
                          double min_sep, double max_sep, int nbins, double *xi_out);

/* ---- vector 2-point correlation (E/B diagnostics): the pair accumulation of treegp/utils.py:5-74 ----
 * All pairs i < j; log|p_j - p_i| binned on `edges` (nbins + 1 increasing values = the uniform
 * edges np.histogram builds for (bins, range); last edge inclusive).  acc_out is (7, nbins):
 * pairs, sum log r, sum (dx_i dx_j + dy_i dy_j), Re and Im of sum v_i v_j (v = dx + i dy), Re and
 * Im of sum v_i v_j conj(d)^2/|d|^2.  xi_+ = acc[2]/acc[0], xi_- = acc[5]/acc[0], ...        */
int tgp_vcorr(tgp_ctx *ctx, const double *x, const double *y, const double *dx, const double *dy,
              int64_t n, const double *edges, int nbins, double *acc_out);

/* ---- mean function: uniform mean of the k nearest neighbours of each X in the table (X0, y0) -----
 * replaces KNeighborsRegressor(n_neighbors=k).fit(X0, y0).predict(X) at treegp/gp_interp.py:236-238.
 * k in 1..8 or 16.                                                                            */
int tgp_knn_mean(tgp_ctx *ctx, const double *X0, const double *y0, int64_t n0, const double *X, int64_t m,
                 int k, double *out);

/* ---- building the mean function ("meanify"): binned 2-D statistic of scattered values ------------
 * replaces scipy.stats.binned_statistic_2d(u, v, ., bins=[u_edges, v_edges], statistic=.) at
 * treegp/meanify.py:76-107, with scipy's bin numbering (np.digitize, last edge inclusive).
 * stat MEAN: sum/count; MEDIAN: scipy's median of the bin; WEIGHTED: w = 1/err^2,
 * average = sum(w p)/sum(w), wrms = sqrt((sum(w p p) - 2 avg sum(w p) + avg^2 sum(w))/sum(w)).
 * Outputs are (nu_edges-1, nv_edges-1) row-major (u index first, as scipy returns them; the
 * reference transposes afterwards); empty bins are NaN.  wrms (0 unless WEIGHTED) and count
 * (points per bin; sum of weights for WEIGHTED) may be NULL.                                  */
#define TGP_STAT_MEAN 0
#define TGP_STAT_MEDIAN 1
#define TGP_STAT_WEIGHTED 2
int tgp_binned_stat_2d(tgp_ctx *ctx, const double *u, const double *v, const double *val, const double *err,
                       int64_t n, const double *u_edges, int nu_edges, const double *v_edges, int nv_edges,
                       int stat, double *average, double *wrms, double *count);

/* ---- device-resident tier (bench / multi-GPU): same maths, d_ pointers, ctx stream --------*/
int tgp_dev_alloc(tgp_ctx *ctx, int64_t bytes, void **d_out);
int tgp_dev_free(tgp_ctx *ctx, void *d_ptr);
int tgp_h2d(tgp_ctx *ctx, void *d_dst, const void *src, int64_t bytes);
int tgp_d2h(tgp_ctx *ctx, void *dst, const void *d_src, int64_t bytes);
int tgp_sync(tgp_ctx *ctx);
void *tgp_stream(tgp_ctx *ctx);          /* the hipStream_t every kernel is launched on */
int tgp_mem_info(tgp_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);   /* device memory of the context's GPU */

/* d_X (n,2), d_y, d_yerr, d_alpha (n): resident inputs/outputs */
int tgp_d_gp_solve(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n,
                   const double *d_y, const double *d_yerr, double *d_alpha, double *logdet,
                   double *ydota, tgp_factor **keep);
int tgp_d_gp_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n,
                     const double *d_alpha, const double *d_Xs, int64_t m, double *d_ys);
/* d_K: dense (n, n) row-major on the device (lower triangle read) */
int tgp_d_gp_solve_dense(tgp_ctx *ctx, const double *d_K, int64_t n, const double *d_y,
                         const double *d_yerr, double *d_alpha, double *logdet, double *ydota,
                         tgp_factor **keep);

/* ---- factor storage: packed lower panels ("panel-major") -----------------------------------
 * Np = n rounded up to 256.  Panel p (p = 0 .. Np/256-1) holds rows 256p .. Np-1 of columns
 * 256p .. 256p+255, row-major with leading dimension 256; panels are stored back to back.
 * tgp_panel_elems(Np) doubles in total.  Element (i, j), i >= 256*(j/256):
 *   off = tgp_panel_off(j/256, Np) + (i - 256*(j/256))*256 + j%256                            */
int64_t tgp_panel_off(int64_t p, int64_t Np);
int64_t tgp_panel_elems(int64_t Np);
int64_t tgp_padded_n(int64_t n);

/* building blocks, exposed for parity tests, profiling and the multi-GPU driver */
int tgp_d_kbuild_lower(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n,
                       const double *d_yerr, double *d_A);
/* in-place Cholesky of the packed lower matrix; d_W receives (Np/128) inverted 128x128
 * diagonal blocks of L.  Returns 0 or the 1-based index of the first non-positive pivot. */
int tgp_d_potrf(tgp_ctx *ctx, double *d_A, int64_t Np, double *d_W);
/* d_b (Np) <- L^-T L^-1 d_b */
int tgp_d_potrs(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b);
/* d_B (nrhs, Np) row-major: every row <- L^-T L^-1 row */
int tgp_d_potrs_multi(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_B, int nrhs);
/* unpack the lower triangle into a dense (n, n) row-major host matrix (upper part zero) */
int tgp_d_unpack_lower(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *out);
/* ---- multi-GPU tier (one process per GPU; driver: treegp_amd/dist.py) ------------------------
 * Row-block-cyclic over 256-row blocks with the deal reflected every G blocks: round q = b / G gives
 * rank g block q G + g when q is even and q G + (G-1-g) when q is odd, so that the lower triangle's
 * work (block row b carries b + 1 block columns) is even over the ranks.  A rank's local index of block
 * b is b / G; an all-gathered panel is [rank][cmax][256][256] with cmax = the most blocks any rank holds.
 * A rank's share of panel p (blocks b >= p) is stored like a single-GPU panel at element offset
 * tgp_dist_panel_off(p).  h_loff / d_loff are host / device copies of those offsets (Np/256+1
 * entries).  These calls only enqueue work on the context stream (tgp_set_stream points it at the
 * caller's stream so that RCCL collectives order with it); they replace, per panel, the same
 * reference lines as tgp_gp_solve (treegp/gp_interp.py:180-182).                              */
int tgp_set_stream(tgp_ctx *ctx, void *hip_stream);      /* launch on the caller's stream (NULL = default stream) */
int tgp_reset_stream(tgp_ctx *ctx);                      /* back to the context's own stream */
int tgp_set_side_stream(tgp_ctx *ctx, void *hip_stream); /* lend the look-ahead stream of the context's own schedules */
int64_t tgp_dist_panel_rows(int64_t p, int64_t Np, int G, int g);
int64_t tgp_dist_panel_off(int64_t p, int64_t Np, int G, int g);
int64_t tgp_dist_local_elems(int64_t Np, int G, int g);
int tgp_dd_kbuild(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_yerr,
                  double *d_Aloc, const int64_t *d_loff, int G, int g);
/* owner of panel kpanel: factor the diagonal block, pack [L_kk(256x256) | W0 | W1] = 98304 doubles */
int tgp_dd_factor_diag(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, int64_t Np, int kpanel, int G, int g,
                       double *d_W, double *d_bcast);
/* every rank, after the broadcast of d_bcast: solve the local rows of panel kpanel (16-row slices while the rank holds at
 * most 24 row tiles, 128-row tiles above); a receiver's d_W gets the panel's inverted blocks from d_bcast by the same grid */
int tgp_dd_trsm(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, int64_t Np, int kpanel, int G, int g,
                double *d_W, const double *d_bcast);
/* every rank, after the all-gather of the panel ([rank][cmax][256][256]): update the local block rows,
 * trailing 128-tile columns [col_lo, col_hi) only (col_hi < 0: to the end) -- the first two columns are
 * the next panel, updated first so that its factorisation overlaps the rest (look-ahead)          */
int tgp_dd_update(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                  const double *d_gathered, int cmax, int col_lo, int col_hi);
/* the same after the PAIR of panels (kpanel, kpanel+1) in one pass of depth 512: gathered0 / gathered1 are
 * the all-gathered panels kpanel (blocks > kpanel, cmax0 per rank) and kpanel+1 (blocks > kpanel+1, cmax1);
 * tile columns col_lo..col_hi count from block kpanel+2.                                              */
int tgp_dd_update2(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                   const double *d_gathered0, int cmax0, const double *d_gathered1, int cmax1, int col_lo,
                   int col_hi);
/* general form: after the GROUP of nseg (1..4) consecutive panels kpanel .. kpanel+nseg-1 in one pass of depth
 * 256 nseg.  d_gathered[s] / cmax[s] (host arrays) describe the all-gathered panel kpanel+s (blocks > kpanel+s);
 * tile columns col_lo..col_hi count from block kpanel+nseg.  nseg = 1 and 2 are tgp_dd_update / tgp_dd_update2.   */
int tgp_dd_update_group(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                        int nseg, const double *const *d_gathered, const int *cmax, int col_lo, int col_hi);
/* Panel chain with the panel exchange off it (dist.py, TGP_DIST_CHAIN_BCAST; the seam is the same treegp/gp_interp.py:180-182):
 * block b = kgroup + j of a group (1 <= j <= 3) against the j earlier panels of the group in one pass of depth 256 j, this rank's
 * rows of b's two tile columns.  Rows: the rank's own, where they are stored (h_loff: the host copy of d_loff).  Columns: block
 * b's rows of those panels, d_ext[s] (256 x 256 doubles, s = panel - kgroup), as appended by b's owner to the broadcast of its
 * diagonal block; NULL on the owner.  Reads no all-gathered panel.                                                         */
int tgp_dd_strip_left(tgp_ctx *ctx, double *d_Aloc, const int64_t *h_loff, const int64_t *d_loff, int64_t Np, int kgroup, int b,
                      int G, int g, const double *d_ext);
/* The bulk part of the update as a persistent grid that keeps `nres` (1..3) compute units per shader engine and XCD clear for
 * the panel chain (diagonal block, local solves, strips) that runs beside it on another stream; for the steps where this
 * rank's share of the bulk is shorter than the chain.  tgp_dd_queue_reset: once per factorisation, on the launching context;
 * tgp_dd_set_exclusive(chain ctx, 1): its diagonal blocks then take a compute unit of their own (44 us instead of ~170 us
 * beside bulk waves) -- only while such a launch keeps units clear.                                                        */
int tgp_dd_queue_reset(tgp_ctx *ctx);
int tgp_dd_set_exclusive(tgp_ctx *ctx, int on);
int tgp_dd_update_group_queued(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g,
                               int nseg, const double *const *d_gathered, const int *cmax, int col_lo, int col_hi, int nres);
/* the whole update after a group in ONE launch, the tile columns [0, head_cols) first; the launch publishes a flag when
 * they are done and tgp_dd_wait_head parks the stream of `waiter` (the panel chain's context) on it.  Only where
 * tgp_handoff_mode(ctx) == 1 (hand-offs by stream wait-value; 2 = events: split the update in two calls instead).   */
int tgp_dd_update_group_fused(tgp_ctx *ctx, double *d_Aloc, const int64_t *d_loff, int64_t Np, int kpanel, int G, int g, int nseg,
                              const double *const *d_gathered, const int *cmax, int head_cols, int nres);
int tgp_dd_wait_head(tgp_ctx *ctx, tgp_ctx *waiter);
/* replicated finish of the factorisation: after ONE all-gather of every rank's share of the trailing matrix from panel k0 on
 * (d_gathered: [G][stride] doubles, a rank's region exactly as it stores it), assemble the packed lower matrix of order
 * Np - 256 k0 at d_tail (the replicated factor's tail, or a buffer of tgp_panel_elems(Np - 256 k0)); the caller factors it
 * with tgp_d_potrf and tgp_dd_tail_scatter copies this rank's blocks of the result back into its share.               */
int tgp_dd_tail_assemble(tgp_ctx *ctx, const double *d_gathered, int64_t stride, int64_t Np, int k0, int G, double *d_tail);
int tgp_dd_tail_scatter(tgp_ctx *ctx, const double *d_tail, int64_t Np, int k0, int G, int g, double *d_Aloc, const int64_t *d_loff);
/* how this context's streams hand over to each other: 1 = flags + hipStreamWaitValue32, 2 = events (chosen by a first-use
 * trial per device, or TGP_SYNC_EVENTS=1 / 0) */
int tgp_handoff_mode(tgp_ctx *ctx);
/* Replicated factor for the solves: store panel kpanel into d_Afull, a full single-GPU packed matrix
 * (tgp_panel_elems(Np) doubles) kept on every rank -- its 256x256 diagonal block from the broadcast buffer
 * (d_bcast, may be NULL) and/or the rows below it from the all-gathered panel (d_gathered, may be NULL).
 * After the last panel, tgp_d_potrs(ctx, d_Afull, d_W, Np, rhs) solves on every rank without communication. */
int tgp_dd_keep_panel(tgp_ctx *ctx, double *d_Afull, int64_t Np, int kpanel, int G, const double *d_bcast,
                      const double *d_gathered, int cmax);
/* block-row-cyclic triangular solves (scipy cho_solve, gp_interp.py:182) */
int tgp_dd_fwd_diag(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int kb, const double *d_W, double *d_yk);
int tgp_dd_fwd_update(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int64_t Np, int kb, int G, int g,
                      const double *d_zk, double *d_yloc);
int tgp_dd_bwd_partial(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int64_t Np, int kb, int G, int g,
                       const double *d_aloc, double *d_s);
int tgp_dd_bwd_diag(tgp_ctx *ctx, const double *d_Aloc, const int64_t *h_loff, int kb, const double *d_W, double *d_ak,
                    const double *d_s /* subtracted from a_k first; may be NULL */);
int tgp_dd_logdet_local(tgp_ctx *ctx, const double *d_Aloc, const int64_t *d_loff, int64_t Np, int64_t n, int G, int g,
                        double *d_out);
int tgp_dd_info(tgp_ctx *ctx, int reset);                /* first non-PD pivot since the last reset */

/* measurement hook: `reps` launches of the depth-512 trailing update over a whole packed Np x Np
 * matrix (contents arbitrary), timed with HIP events on the context's stream */
int tgp_debug_syrk_loop(tgp_ctx *ctx, double *d_A, int64_t Np, int reps, double *ms_per_launch,
                        double *flops_per_launch);
/* test hook: tile enumeration of the trailing update over T x T 128-tiles; fills (ti, tj)
 * for every block id (-1 = empty slot) and returns the grid size, or -1 if cap is too small */
int tgp_debug_tilemap(int64_t T, int32_t *ti, int32_t *tj, int64_t cap);

#ifdef __cplusplus
}
#endif
#endif /* TGP_H */
