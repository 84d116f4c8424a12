// C-ABI glue of libtgp.so (include/tgp.h): context, staging of host buffers, phase timings.
#include "tgp_internal.h"

#include <chrono>

namespace {
struct Arena {               // bump allocator over the ctx staging buffer (sized up front)
    char *base;
    size_t off = 0;
    template <typename T> T *take(size_t count) {
        T *p = reinterpret_cast<T *>(base + off);
        off += (count * sizeof(T) + 255) / 256 * 256;
        return p;
    }
};
inline size_t rup(size_t b) { return (b + 255) / 256 * 256; }

struct Staging {
    void *buf = nullptr;
    size_t bytes = 0;
};
Staging g_dummy;
}  // namespace

// per-ctx staging lives in a side table keyed by ctx (keeps tgp_ctx POD-ish)
struct tgp_ctx_ext {
    Staging io;
    Staging hio;                 // pinned host mirror of the front of `io` (same offsets), see h2d / d2h_sync
    Staging hpin;                // pinned host scratch (tgp_ensure_pinned)
    double *A_cache = nullptr;   // packed lower panels, reused across solves of the same size
    double *W_cache = nullptr;
    int64_t cache_Np = 0;
};
static tgp_ctx_ext *ext_of(tgp_ctx *ctx);

struct tgp_ctx_full : tgp_ctx {
    tgp_ctx_ext ext;
};
static tgp_ctx_ext *ext_of(tgp_ctx *ctx) { return &static_cast<tgp_ctx_full *>(ctx)->ext; }

int tgp_ensure_scratch(tgp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return 0;
    if (ctx->scratch) TGP_HIP(hipFree(ctx->scratch));
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    TGP_HIP(hipMalloc(&ctx->scratch, bytes));
    ctx->scratch_bytes = bytes;
    return 0;
}

// the look-ahead stream is created on first use: contexts that never factorise with look-ahead (small problems,
// tgp_set_lookahead(ctx, 0), contexts whose stream is set by the caller and only carry kernels) do not take a slot in
// the runtime's rotation of streams over its few hardware queues
int tgp_ensure_side_stream(tgp_ctx *ctx) {
    if (ctx->side_stream) return 0;
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    TGP_HIP(hipStreamCreateWithPriority(&ctx->side_stream, hipStreamNonBlocking, hi));
    return 0;
}

int tgp_ensure_scratch2(tgp_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->scratch2_bytes) return 0;
    if (ctx->scratch2) TGP_HIP(hipFree(ctx->scratch2));
    ctx->scratch2 = nullptr;
    ctx->scratch2_bytes = 0;
    TGP_HIP(hipMalloc(&ctx->scratch2, bytes));
    ctx->scratch2_bytes = bytes;
    return 0;
}

static int ensure_io(tgp_ctx *ctx, size_t bytes) {
    Staging &s = ext_of(ctx)->io;
    if (bytes <= s.bytes) return 0;
    if (s.buf) TGP_HIP(hipFree(s.buf));
    s.buf = nullptr;
    s.bytes = 0;
    TGP_HIP(hipMalloc(&s.buf, bytes));
    s.bytes = bytes;
    return 0;
}

static int ensure_factor_cache(tgp_ctx *ctx, int64_t Np) {
    tgp_ctx_ext *e = ext_of(ctx);
    if (e->A_cache && e->cache_Np == Np) return 0;
    if (e->A_cache) TGP_HIP(hipFree(e->A_cache));
    if (e->W_cache) TGP_HIP(hipFree(e->W_cache));
    e->A_cache = e->W_cache = nullptr;
    e->cache_Np = 0;
    TGP_HIP(hipMalloc((void **)&e->A_cache, (size_t)tgp_panel_elems(Np) * sizeof(double)));
    TGP_HIP(hipMalloc((void **)&e->W_cache, (size_t)Np * TGP_TB * sizeof(double)));
    e->cache_Np = Np;
    return 0;
}

// Host-boundary copies of the hot calls (tgp_gp_solve, tgp_gp_predict) go through a pinned mirror of the device staging arena.
// A pageable source or destination makes the runtime pin the caller's pages on the fly and unpin them later, asynchronously:
// at the headline size that work (4 MB of query points in, 2 MB of predictions out) was still going on when the NEXT call
// arrived -- 10 - 28 ms during which the device counts as busy (hipDeviceSynchronize on entry took that long) and which showed
// up as "host tax" of every API pass but the first (bench.py api_route; found with TGP_HOST_PHASES=1).  The mirror is capped
// at 256 MB; larger transfers (dense kernel matrices) take the direct path.
static int ensure_hio(tgp_ctx *ctx, size_t bytes) {
    constexpr size_t cap = (size_t)256 << 20;
    Staging &s = ext_of(ctx)->hio;
    if (bytes > cap) bytes = cap;
    if (bytes <= s.bytes) return 0;
    if (s.buf) TGP_HIP(hipHostFree(s.buf));
    s.buf = nullptr;
    s.bytes = 0;
    TGP_HIP(hipHostMalloc(&s.buf, bytes, hipHostMallocDefault));
    s.bytes = bytes;
    return 0;
}
int tgp_ensure_pinned(tgp_ctx *ctx, size_t bytes, void **out) {
    Staging &s = ext_of(ctx)->hpin;
    if (bytes > s.bytes) {
        if (s.buf) TGP_HIP(hipHostFree(s.buf));
        s.buf = nullptr;
        s.bytes = 0;
        TGP_HIP(hipHostMalloc(&s.buf, bytes, hipHostMallocDefault));
        s.bytes = bytes;
    }
    *out = s.buf;
    return 0;
}
static void *hio_mirror(tgp_ctx *ctx, const void *d_ptr, size_t bytes) {
    tgp_ctx_ext *e = ext_of(ctx);
    const size_t off = (size_t)((const char *)d_ptr - (const char *)e->io.buf);
    return (e->hio.buf && (const char *)d_ptr >= (const char *)e->io.buf && off + bytes <= e->hio.bytes) ? (char *)e->hio.buf + off : nullptr;
}
static hipError_t h2d(tgp_ctx *ctx, void *d_dst, const void *src, size_t bytes) {
    if (void *m = hio_mirror(ctx, d_dst, bytes)) {
        memcpy(m, src, bytes);
        src = m;
    }
    return hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, ctx->stream);
}
// device -> caller's array, then the stream is synchronised (the mirror is free again for the next call)
static hipError_t d2h_sync(tgp_ctx *ctx, void *dst, const void *d_src, size_t bytes) {
    void *m = hio_mirror(ctx, d_src, bytes);
    hipError_t e = hipMemcpyAsync(m ? m : dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess && m) memcpy(dst, m, bytes);
    return e;
}

int tgp_ensure_io(tgp_ctx *ctx, size_t bytes) { return ensure_io(ctx, bytes); }
void *tgp_io_buffer(tgp_ctx *ctx) { return ext_of(ctx)->io.buf; }
// a kept factor goes back to being the context's factor cache (the next tgp_d_gp_solve of the same size reuses its memory)
void tgp_factor_release_to_cache(tgp_ctx *ctx, tgp_factor *f) {
    if (!f) return;
    if (f->borrowed) { tgp_factor_free(ctx, f); return; }
    tgp_ctx_ext *e = ext_of(ctx);
    if (e->A_cache) (void)hipFree(e->A_cache);
    if (e->W_cache) (void)hipFree(e->W_cache);
    e->A_cache = f->d_A;
    e->W_cache = f->d_W;
    e->cache_Np = f->Np;
    // its slabs become the context's (the next solve that keeps a factor of this size takes them back, factor_and_solve):
    // no hipMalloc / hipFree -- a device-wide synchronisation each -- per likelihood evaluation of a gradient-driven fit
    if (f->d_slabs && !ctx->vslab) {
        ctx->vslab = f->d_slabs;
        ctx->vslab_bytes = vslab_bytes(f->Np, f->slab_S);
    } else if (f->d_slabs) {
        (void)hipFree(f->d_slabs);
    }
    if (f->d_slabs2) (void)hipFree(f->d_slabs2);
    delete f;
}

std::atomic<int> tgp_solves_in_flight{0};
namespace {
struct SolveInFlight {
    bool alone;
    SolveInFlight() : alone(tgp_solves_in_flight.fetch_add(1) == 0) {}
    ~SolveInFlight() { tgp_solves_in_flight.fetch_sub(1); }
};
}  // namespace

extern "C" {

const char *tgp_version(void) { return "treegp_amd libtgp 0.1 (gfx950)"; }

int tgp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int tgp_init(const int *devices, int ndev, tgp_ctx **out) {
    if (!out || ndev != 1 || !devices) return -1;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return -3;   // no HIP device: fail loudly
    if (devices[0] < 0 || devices[0] >= n) return -1;
    tgp_ctx_full *ctx = new tgp_ctx_full();
    ctx->device = devices[0];
    if (hipSetDevice(ctx->device) != hipSuccess) { delete ctx; return -2; }
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return -2; }
    ctx->stream = ctx->own_stream;
    for (auto &e : ctx->ev)
        if (hipEventCreate(&e) != hipSuccess) { delete ctx; return -2; }
    if (hipMalloc((void **)&ctx->d_info, 256) != hipSuccess) { delete ctx; return -2; }
    if (hipMalloc((void **)&ctx->d_queue, TGP_NQUEUE * TGP_QUEUE_WORDS * sizeof(unsigned)) != hipSuccess) { delete ctx; return -2; }
    if (hipMemset(ctx->d_info, 0, 256) != hipSuccess) { delete ctx; return -2; }
    if (hipMalloc((void **)&ctx->d_psync, TGP_PSYNC_PANELS * 16 * sizeof(unsigned)) != hipSuccess) { delete ctx; return -2; }
    if (hipMalloc((void **)&ctx->d_flags, 16 * 64) != hipSuccess || hipMemset(ctx->d_flags, 0, 16 * 64) != hipSuccess) { delete ctx; return -2; }
    if (const char *e = getenv("TGP_FLAG_SEQ_START")) {      // test hook: start the hand-off sequence numbers near their wrap-around
        const unsigned v0 = (unsigned)strtoul(e, nullptr, 10);
        unsigned h[16 * 16];
        for (unsigned &x : h) x = v0;
        if (hipMemcpy(ctx->d_flags, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) { delete ctx; return -2; }
        for (unsigned &q : ctx->flag_seq) q = v0;
        ctx->head_count = v0;
    }
    if (hipHostMalloc((void **)&ctx->h_info, 256, hipHostMallocDefault) != hipSuccess) { delete ctx; return -2; }
    if (hipMalloc((void **)&ctx->d_scal, 16 * sizeof(double)) != hipSuccess) { delete ctx; return -2; }
    if (hipHostMalloc((void **)&ctx->h_scal, 16 * sizeof(double), hipHostMallocDefault) != hipSuccess) { delete ctx; return -2; }
    *out = ctx;
    return 0;
}

void tgp_destroy(tgp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    tgp_ctx_ext *e = ext_of(ctx);
    if (e->io.buf) (void)hipFree(e->io.buf);
    if (e->hio.buf) (void)hipHostFree(e->hio.buf);
    if (e->hpin.buf) (void)hipHostFree(e->hpin.buf);
    if (e->A_cache) (void)hipFree(e->A_cache);
    if (e->W_cache) (void)hipFree(e->W_cache);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->scratch2) (void)hipFree(ctx->scratch2);
    if (ctx->vslab) (void)hipFree(ctx->vslab);
    if (ctx->vslab_tt) (void)hipFree(ctx->vslab_tt);
    if (ctx->d_info) (void)hipFree(ctx->d_info);
    if (ctx->d_queue) (void)hipFree(ctx->d_queue);
    if (ctx->d_psync) (void)hipFree(ctx->d_psync);
    if (ctx->d_flags) (void)hipFree(ctx->d_flags);
    if (ctx->h_info) (void)hipHostFree(ctx->h_info);
    if (ctx->d_scal) (void)hipFree(ctx->d_scal);
    if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
    for (auto &ev : ctx->ev)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->ev_slab)
        if (ev) (void)hipEventDestroy(ev);
    for (auto &ev : ctx->prof_events) (void)hipEventDestroy(ev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->side_stream && !ctx->ext_side_stream) (void)hipStreamDestroy(ctx->side_stream);
    delete static_cast<tgp_ctx_full *>(ctx);
}

const char *tgp_last_error(tgp_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

int tgp_last_timings(tgp_ctx *ctx, double *ms, int n) {
    if (!ctx || !ms) return -1;
    for (int i = 0; i < n && i < TGP_NTIMINGS; ++i) ms[i] = ctx->timings[i];
    return 0;
}

int tgp_set_lookahead(tgp_ctx *ctx, int on) {
    if (!ctx) return -1;
    ctx->lookahead = on ? 1 : 0;
    return 0;
}

int tgp_set_profiling(tgp_ctx *ctx, int on) {
    if (!ctx) return -1;
    ctx->profiling = on;
    return 0;
}

int64_t tgp_panel_off(int64_t p, int64_t Np) { return panel_off(p, Np); }
int64_t tgp_panel_elems(int64_t Np) { return panel_off(Np / TGP_PW, Np); }
int64_t tgp_padded_n(int64_t n) { return padded_n(n); }

int tgp_dev_alloc(tgp_ctx *ctx, int64_t bytes, void **d_out) {
    TGP_ARG(bytes > 0 && d_out);
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipMalloc(d_out, (size_t)bytes));
    return 0;
}
int tgp_dev_free(tgp_ctx *ctx, void *d_ptr) {
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    TGP_HIP(hipFree(d_ptr));
    return 0;
}
int tgp_h2d(tgp_ctx *ctx, void *d_dst, const void *src, int64_t bytes) {
    TGP_HIP(hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int tgp_d2h(tgp_ctx *ctx, void *dst, const void *d_src, int64_t bytes) {
    TGP_HIP(hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
int tgp_sync(tgp_ctx *ctx) {
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}
void *tgp_stream(tgp_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int tgp_mem_info(tgp_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes) {
    TGP_ARG(free_bytes && total_bytes);
    TGP_HIP(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    TGP_HIP(hipMemGetInfo(&f, &t));
    *free_bytes = (int64_t)f;
    *total_bytes = (int64_t)t;
    return 0;
}

// ---- building blocks ---------------------------------------------------------------------
int tgp_d_kbuild_lower(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_yerr,
                       double *d_A) {
    TGP_ARG(k && d_X && d_A && n > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    const int64_t Np = padded_n(n);
    TGP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc = launch_kbuild_lower(ctx, k, d_X, n, Np, d_yerr, d_A);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[0] = ms;
    ctx->timings[8] = 8.0 * ((double)Np * (Np + 1) / 2.0) + 16.0 * (double)Np;
    return 0;
}

int tgp_d_potrf(tgp_ctx *ctx, double *d_A, int64_t Np, double *d_W) {
    TGP_ARG(d_A && d_W);
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    int info = launch_potrf(ctx, d_A, Np, d_W);
    if (info < 0) return info;
    TGP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[1] = ms;
    return info;
}

int tgp_d_potrs(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_b) {
    TGP_ARG(d_A && d_W && d_b && Np % TGP_PW == 0);
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc = launch_potrs(ctx, d_A, d_W, Np, d_b);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[2] = ms;
    return 0;
}

int tgp_d_unpack_lower(tgp_ctx *ctx, const double *d_A, int64_t Np, int64_t n, double *out) {
    TGP_ARG(d_A && out && n > 0 && n <= Np);
    TGP_HIP(hipSetDevice(ctx->device));
    int rc = ensure_io(ctx, (size_t)n * n * sizeof(double));
    if (rc) return rc;
    double *d_out = (double *)ext_of(ctx)->io.buf;
    rc = launch_unpack_lower(ctx, d_A, Np, n, d_out);
    if (rc) return rc;
    TGP_HIP(hipMemcpyAsync(out, d_out, (size_t)n * n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- S2 -------------------------------------------------------------------------------------
// What follows the K build, shared by the parametrised kernels (tgp_d_gp_solve) and the caller-built matrix
// (tgp_d_gp_solve_dense): factorise the packed matrix in the context's cache, solve, log-determinant, y . alpha.
// ev[0] / ev[1] bracket the build (recorded by the caller).
static int factor_and_solve(tgp_ctx *ctx, int64_t n, int64_t Np, const double *d_y, double *d_alpha, double *logdet,
                            double *ydota, tgp_factor **keep, bool augmented) {
    hipStream_t st = ctx->stream;
    tgp_ctx_ext *e = ext_of(ctx);
    double *d_A = e->A_cache, *d_W = e->W_cache;
    double *d_b = (double *)ctx->scratch;
    int rc = 0;
    // A kept factor owns the slabs built for its solve (nothing is rebuilt at the first covariance / multi-field / gradient call).
    static const bool both_sweeps = getenv("TGP_CHI2_BOTH_SWEEPS") != nullptr;      // A/B: y . alpha as before
    const bool forward_only = d_alpha == nullptr && !both_sweeps;                   // only |L^-1 y|^2 is wanted
    double *own_slabs = nullptr;          // slabs that will belong to the kept factor
    int own_S = 0;
    auto fail = [&](int code) {
        if (own_slabs) (void)hipFree(own_slabs);
        return code;
    };
    // The solves are queued behind the factorisation without waiting for its verdict (one host round trip less per
    // likelihood evaluation); if a pivot failed they run on a meaningless factor and their result is discarded below.
    static const bool dbg = getenv("TGP_HOST_PHASES") != nullptr;
    const auto h0 = std::chrono::steady_clock::now();
    int info = launch_potrf(ctx, d_A, Np, d_W, /*defer_info=*/true, /*n_data=*/n);
    if (info < 0) return fail(info);
    const auto h1 = std::chrono::steady_clock::now();
    TGP_HIP(hipEventRecord(ctx->ev[2], st));
    double sweeps = 0.0;
    if (info == 0 && augmented) {
        rc = launch_logdet_rowsq(ctx, d_A, Np, n, ctx->d_scal);
        if (rc) return fail(rc);
        if (d_alpha) {
            // alpha wanted too: the augmented row IS L^-1 y, so only the backward sweep is left (slabs built here, in one go).
            // The row's own huge pivot does not enter: its component of the right-hand side is zero.
            int Sa = 0;
            (void)potrs_big_step(Np, &Sa);
            rc = tgp_ensure_scratch2(ctx, (size_t)Np * sizeof(double));
            if (rc) return fail(rc);
            double *d_z = (double *)ctx->scratch2, *slabs = nullptr;
            bool build = false;
            rc = acquire_slabs(ctx, Np, Sa, nullptr, nullptr, &slabs, &build);
            if (rc) return fail(rc);
            rc = launch_vslab_build(ctx, d_A, d_W, Np, Sa, slabs);
            if (rc) return fail(rc);
            rc = launch_extract_row(ctx, d_A, Np, n, d_z);
            if (rc) return fail(rc);
            rc = launch_potrs_big_bwd(ctx, d_A, Np, Sa, slabs, d_b, d_z);
            if (rc) return fail(rc);
            sweeps = 1.0;
            TGP_HIP(hipMemcpyAsync(d_alpha, d_b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        }
        TGP_HIP(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    } else if (info == 0) {
        rc = launch_pad_copy(ctx, d_y, n, Np, d_b);
        if (rc) return fail(rc);
        int S2 = 0;
        if (keep && !own_slabs && potrs_big_step(Np, &S2) && ctx->vslab && ctx->vslab_bytes == vslab_bytes(Np, S2)) {
            own_slabs = (double *)ctx->vslab;      // the context's buffer (e.g. handed back by the previous kept factor of this
            own_S = S2;                            // size, tgp_factor_release_to_cache) moves to the new handle: no hipMalloc
            ctx->vslab = nullptr;
            ctx->vslab_bytes = 0;
        }
        if (own_slabs) {                  // memory in hand (from the context): build here
            rc = launch_vslab_build(ctx, d_A, d_W, Np, own_S, own_slabs);
            if (rc) return fail(rc);
        }
        // without alpha only the quadratic form is wanted: y^T K^-1 y = |L^-1 y|^2, the forward sweep alone
        rc = launch_potrs(ctx, d_A, d_W, Np, d_b, forward_only, keep ? &own_slabs : nullptr, keep ? &own_S : nullptr);
        if (rc) return fail(rc);
        sweeps = forward_only ? 1.0 : 2.0;
        rc = launch_logdet_dot(ctx, d_A, Np, n, forward_only ? d_b : d_y, d_b, ctx->d_scal);
        if (rc) return fail(rc);
        if (d_alpha) TGP_HIP(hipMemcpyAsync(d_alpha, d_b, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
        TGP_HIP(hipMemcpyAsync(ctx->h_scal, ctx->d_scal, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
    }
    TGP_HIP(hipEventRecord(ctx->ev[3], st));
    const auto h2 = std::chrono::steady_clock::now();
    TGP_HIP(hipStreamSynchronize(st));
    if (dbg) {
        const auto h3 = std::chrono::steady_clock::now();
        auto d = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[factor_and_solve] launch_potrf returned after %.2f ms, solves queued after %.2f ms, final sync %.2f ms\n", d(h0, h1), d(h1, h2), d(h2, h3));
    }
    if (info == 0) info = *ctx->h_info;               // the factorisation's verdict (first failing pivot, 1-based)
    if (info < 0) return fail(tgp_potrf_info_rc(ctx, info));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[0] = ms;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[1], ctx->ev[2]));
    ctx->timings[1] = ms;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[2], ctx->ev[3]));
    ctx->timings[2] = ms;
    ctx->timings[8] = 8.0 * ((double)Np * (Np + 1) / 2.0) + 16.0 * (double)Np;
    ctx->timings[10] = sweeps;
    if (info > 0) {
        if (info > n) info = (int)n;     // cannot happen (padding is the identity); defensive
        return fail(info);
    }
    if (logdet) *logdet = ctx->h_scal[0];
    if (ydota) *ydota = ctx->h_scal[1];
    if (keep) {
        tgp_factor *f = new tgp_factor();
        f->n = n;
        f->Np = Np;
        f->d_A = d_A;
        f->d_W = d_W;
        f->d_slabs = own_slabs;
        f->slab_S = own_S;
        e->A_cache = e->W_cache = nullptr;      // ownership moves to the handle
        e->cache_Np = 0;
        *keep = f;
    }
    return 0;
}

// A factorisation that gave up on an in-kernel hand-off (TGP_RC_HANDOFF: other processes' kernels on the same GPU can keep
// panel_mid_kernel's workgroups apart for longer than its bounded wait) has left a half-updated matrix: the context switches that
// kernel off for good and the solve is run once more, K build included.
static int gp_solve_once(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_y,
                         const double *d_yerr, double *d_alpha, double *logdet, double *ydota, tgp_factor **keep);
int tgp_d_gp_solve(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_y,
                   const double *d_yerr, double *d_alpha, double *logdet, double *ydota, tgp_factor **keep) {
    SolveInFlight here;
    ctx->mid_allowed = here.alone ? 1 : 0;
    int rc = gp_solve_once(ctx, k, d_X, n, d_y, d_yerr, d_alpha, logdet, ydota, keep);
    ctx->mid_allowed = 0;
    if (rc == TGP_RC_HANDOFF && !ctx->mid_off) {
        ctx->mid_off = 1;
        rc = gp_solve_once(ctx, k, d_X, n, d_y, d_yerr, d_alpha, logdet, ydota, keep);
    }
    return rc;
}
static int gp_solve_once(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_y,
                         const double *d_yerr, double *d_alpha, double *logdet, double *ydota, tgp_factor **keep) {
    TGP_ARG(k && d_X && d_y && n > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t Np = padded_n(n);
    int rc = ensure_factor_cache(ctx, Np);
    if (rc) return rc;
    double *d_A = ext_of(ctx)->A_cache;
    rc = tgp_ensure_scratch(ctx, (size_t)Np * sizeof(double));
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    rc = launch_kbuild_lower(ctx, k, d_X, n, Np, d_yerr, d_A);
    if (rc) return rc;
    // Likelihood only (no alpha, factor not kept) and a padding row to spare: y rides along as row Np-1 of the matrix and
    // comes out of the factorisation as L^-1 y -- no triangular sweep (trsv.hip: augment_rhs_kernel)
    // With alpha wanted as well (round 3; not for kept factors, which must stay clean), the row's content is the right-hand
    // side of the backward sweep: the forward sweep disappears for every n that is not a multiple of 256 (big-step sweeps only).
    static const bool no_augment = getenv("TGP_NO_AUGMENT") != nullptr || getenv("TGP_CHI2_BOTH_SWEEPS") != nullptr;
    static const bool no_augment_alpha = getenv("TGP_NO_AUGMENT_ALPHA") != nullptr;
    int S_unused = 0;
    const bool augmented = !no_augment && keep == nullptr && n < Np &&
                           (d_alpha == nullptr || (!no_augment_alpha && potrs_big_step(Np, &S_unused)));
    if (augmented) {
        rc = launch_augment_rhs(ctx, d_A, Np, n, d_y);
        if (rc) return rc;
    }
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    return factor_and_solve(ctx, n, Np, d_y, d_alpha, logdet, ydota, keep, augmented);
}

// The same for a matrix the CALLER evaluated: any scikit-learn kernel tree that tgp_kernel cannot describe (Sum,
// WhiteKernel, Matern, ...; treegp/kernels.py:17-59 evals any of them and gp_interp.py:177-183 works with what comes back).
// d_K: dense (n, n) row-major on the device, lower triangle read; d_yerr^2 (may be NULL) is added to the diagonal.
static int gp_solve_dense_once(tgp_ctx *ctx, const double *d_K, int64_t n, const double *d_y, const double *d_yerr,
                               double *d_alpha, double *logdet, double *ydota, tgp_factor **keep);
int tgp_d_gp_solve_dense(tgp_ctx *ctx, const double *d_K, int64_t n, const double *d_y, const double *d_yerr,
                         double *d_alpha, double *logdet, double *ydota, tgp_factor **keep) {
    SolveInFlight here;
    ctx->mid_allowed = here.alone ? 1 : 0;
    int rc = gp_solve_dense_once(ctx, d_K, n, d_y, d_yerr, d_alpha, logdet, ydota, keep);
    ctx->mid_allowed = 0;
    if (rc == TGP_RC_HANDOFF && !ctx->mid_off) {
        ctx->mid_off = 1;
        rc = gp_solve_dense_once(ctx, d_K, n, d_y, d_yerr, d_alpha, logdet, ydota, keep);
    }
    return rc;
}
static int gp_solve_dense_once(tgp_ctx *ctx, const double *d_K, int64_t n, const double *d_y, const double *d_yerr,
                               double *d_alpha, double *logdet, double *ydota, tgp_factor **keep) {
    TGP_ARG(d_K && d_y && n > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t Np = padded_n(n);
    int rc = ensure_factor_cache(ctx, Np);
    if (rc) return rc;
    rc = tgp_ensure_scratch(ctx, (size_t)Np * sizeof(double));
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    rc = launch_pack_lower(ctx, d_K, n, Np, d_yerr, ext_of(ctx)->A_cache);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    return factor_and_solve(ctx, n, Np, d_y, d_alpha, logdet, ydota, keep, false);
}

int tgp_gp_solve_dense(tgp_ctx *ctx, const double *K, int64_t n, const double *y, const double *yerr, double *alpha,
                       double *logdet, double *ydota, tgp_factor **keep) {
    TGP_ARG(K && y && n > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = ensure_io(ctx, rup((size_t)n * n * 8) + 3 * rup(n * 8));
    if (rc) return rc;
    Arena ar{(char *)ext_of(ctx)->io.buf};
    double *d_K = ar.take<double>((size_t)n * n), *d_y = ar.take<double>(n), *d_e = ar.take<double>(n), *d_a = ar.take<double>(n);
    TGP_HIP(hipMemcpyAsync(d_K, K, (size_t)n * n * 8, hipMemcpyHostToDevice, st));
    TGP_HIP(hipMemcpyAsync(d_y, y, n * 8, hipMemcpyHostToDevice, st));
    if (yerr) TGP_HIP(hipMemcpyAsync(d_e, yerr, n * 8, hipMemcpyHostToDevice, st));
    rc = tgp_d_gp_solve_dense(ctx, d_K, n, d_y, yerr ? d_e : nullptr, alpha ? d_a : nullptr, logdet, ydota, keep);
    if (rc) return rc;
    if (alpha) {
        TGP_HIP(hipMemcpyAsync(alpha, d_a, n * 8, hipMemcpyDeviceToHost, st));
        TGP_HIP(hipStreamSynchronize(st));
    }
    return 0;
}

// ---- several right-hand sides against one kept factor (the Piff pattern, treegp/README.rst:28: one GP per PSF parameter,
// all sharing the star positions) -------------------------------------------------------------------------------------
// d_B: (nrhs, Np) row-major on the device, each row one padded right-hand side, solved in place
int tgp_d_potrs_multi(tgp_ctx *ctx, const double *d_A, const double *d_W, int64_t Np, double *d_B, int nrhs) {
    TGP_ARG(d_A && d_W && d_B && nrhs > 0 && Np % TGP_PW == 0);
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc = launch_potrs_multi(ctx, d_A, d_W, Np, d_B, nrhs, nullptr);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[2] = ms;
    return 0;
}

// B, Xout: (nrhs, n) row-major host; Xout[v] = (K + D)^-1 B[v] with the factor kept by tgp_gp_solve / tgp_gp_solve_dense
int tgp_factor_solve(tgp_ctx *ctx, tgp_factor *f, const double *B, int nrhs, double *Xout) {
    TGP_ARG(f && B && Xout && nrhs > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const int64_t n = f->n, Np = f->Np;
    int rc = ensure_io(ctx, rup((size_t)nrhs * Np * 8));
    if (rc) return rc;
    double *d_B = (double *)ext_of(ctx)->io.buf;
    TGP_HIP(hipMemsetAsync(d_B, 0, (size_t)nrhs * Np * 8, st));
    TGP_HIP(hipMemcpy2DAsync(d_B, (size_t)Np * 8, B, (size_t)n * 8, (size_t)n * 8, (size_t)nrhs, hipMemcpyHostToDevice, st));
    TGP_HIP(hipEventRecord(ctx->ev[0], st));
    rc = launch_potrs_multi(ctx, f->d_A, f->d_W, Np, d_B, nrhs, &f->d_slabs, &f->slab_S);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], st));
    TGP_HIP(hipMemcpy2DAsync(Xout, (size_t)n * 8, d_B, (size_t)Np * 8, (size_t)n * 8, (size_t)nrhs, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[2] = ms;
    return 0;
}

// the context gives back what it holds between calls: the factor cache of its last solve (17 GB at N = 65 536), inverse slabs,
// scratch -- for a process about to allocate elsewhere on the same GPU (the multi-GPU engine before its replicated factor)
int tgp_release_caches(tgp_ctx *ctx) {
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    tgp_ctx_ext *e = ext_of(ctx);
    if (e->A_cache) (void)hipFree(e->A_cache);
    if (e->W_cache) (void)hipFree(e->W_cache);
    e->A_cache = e->W_cache = nullptr;
    e->cache_Np = 0;
    if (ctx->vslab) (void)hipFree(ctx->vslab);
    ctx->vslab = nullptr;
    ctx->vslab_bytes = 0;
    if (ctx->vslab_tt) (void)hipFree(ctx->vslab_tt);
    ctx->vslab_tt = nullptr;
    ctx->vslab_tt_bytes = 0;
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    return 0;
}

int tgp_gp_solve(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n, const double *y,
                 const double *yerr, double *alpha, double *logdet, double *ydota, tgp_factor **keep) {
    TGP_ARG(k && X && y && n > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = ensure_io(ctx, rup(2 * n * 8) + 3 * rup(n * 8));
    if (rc) return rc;
    rc = ensure_hio(ctx, rup(2 * n * 8) + 3 * rup(n * 8));
    if (rc) return rc;
    Arena ar{(char *)ext_of(ctx)->io.buf};
    double *d_X = ar.take<double>(2 * n), *d_y = ar.take<double>(n), *d_e = ar.take<double>(n),
           *d_a = ar.take<double>(n);
    const bool dbg = getenv("TGP_HOST_PHASES") != nullptr;      // development: where the host-boundary call spends its wall time
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
        return std::chrono::duration<double, std::milli>(b - a).count();
    };
    const auto t0 = now();
    hipEvent_t evA = nullptr;
    if (dbg) {
        const auto p0 = now();
        TGP_HIP(hipStreamSynchronize(st));
        const auto p1 = now();
        TGP_HIP(hipDeviceSynchronize());
        fprintf(stderr, "[tgp_gp_solve] on entry: stream had %.2f ms of work left, the device %.2f ms more\n", ms(p0, p1), ms(p1, now()));
        TGP_HIP(hipEventCreate(&evA));
        TGP_HIP(hipEventRecord(evA, st));
    }
    TGP_HIP(h2d(ctx, d_X, X, 2 * n * 8));
    TGP_HIP(h2d(ctx, d_y, y, n * 8));
    if (yerr) TGP_HIP(h2d(ctx, d_e, yerr, n * 8));
    const auto t1 = now();
    rc = tgp_d_gp_solve(ctx, k, d_X, n, d_y, yerr ? d_e : nullptr, alpha ? d_a : nullptr, logdet, ydota, keep);
    if (rc) return rc;
    const auto t2 = now();
    if (alpha) TGP_HIP(d2h_sync(ctx, alpha, d_a, n * 8));
    if (dbg) {
        float lead = 0.f;      // on the device: from the stream reaching this call to the K build's start (the copies in between)
        (void)hipEventElapsedTime(&lead, evA, ctx->ev[0]);
        (void)hipEventDestroy(evA);
        fprintf(stderr, "[tgp_gp_solve n=%ld] H2D %.2f ms on the host / %.2f ms on the stream, tgp_d_gp_solve %.2f ms (device phases %.2f), D2H %.2f ms\n",
                (long)n, ms(t0, t1), lead, ms(t1, t2), ctx->timings[0] + ctx->timings[1] + ctx->timings[2], ms(t2, now()));
    }
    return 0;
}

void tgp_factor_free(tgp_ctx *ctx, tgp_factor *f) {
    if (!f) return;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
    }
    if (f->d_A && !f->borrowed) (void)hipFree(f->d_A);
    if (f->d_W && !f->borrowed) (void)hipFree(f->d_W);
    if (f->d_slabs) (void)hipFree(f->d_slabs);
    if (f->d_slabs2) (void)hipFree(f->d_slabs2);
    delete f;
}

// The handle is given up but its device memory stays with the context as the factor cache of the next solve of this size
// (what tgp_d_gp_solve allocates otherwise): a GPInterpolation that is dropped and followed by another of the same size --
// a refit, the next exposure -- then costs no hipFree + hipMalloc of the packed matrix (17 GB at N = 65 536: ~20 ms of the
// following solve went to it, bench.py api_route before round 4's end).  Memory held this way is freed by tgp_destroy, by a
// solve of another size, or by tgp_factor_free on a handle.  A borrowed handle is simply released.
void tgp_factor_release(tgp_ctx *ctx, tgp_factor *f) {
    if (!f) return;
    if (!ctx) { tgp_factor_free(ctx, f); return; }
    (void)hipSetDevice(ctx->device);
    tgp_factor_release_to_cache(ctx, f);
}

// A handle on a factor that lives in the CALLER's device memory (packed panels d_A + inverted diagonal blocks d_W, as
// tgp_d_potrf leaves them): what the multi-GPU driver's replicated factor is.  Everything that takes a tgp_factor
// (tgp_factor_solve, tgp_gp_predict_cov, tgp_gp_loglik_grad) then works on it; tgp_factor_free releases the handle and
// the slabs it built, never d_A / d_W.
int tgp_factor_borrow(tgp_ctx *ctx, double *d_A, double *d_W, int64_t n, tgp_factor **out) {
    TGP_ARG(d_A && d_W && out && n > 0);
    tgp_factor *f = new tgp_factor();
    f->n = n;
    f->Np = padded_n(n);
    f->d_A = d_A;
    f->d_W = d_W;
    f->borrowed = true;
    *out = f;
    return 0;
}

// ---- S3 ---------------------------------------------------------------------------------------
int tgp_d_gp_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_alpha,
                     const double *d_Xs, int64_t m, double *d_ys) {
    TGP_ARG(k && d_X && d_alpha && d_Xs && d_ys);
    TGP_HIP(hipSetDevice(ctx->device));
    TGP_HIP(hipEventRecord(ctx->ev[0], ctx->stream));
    int rc = launch_predict(ctx, k, d_X, n, d_alpha, d_Xs, m, d_ys);
    if (rc) return rc;
    TGP_HIP(hipEventRecord(ctx->ev[1], ctx->stream));
    TGP_HIP(hipStreamSynchronize(ctx->stream));
    float ms = 0.f;
    TGP_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[1]));
    ctx->timings[3] = ms;
    return 0;
}

int tgp_gp_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n, const double *alpha,
                   const double *Xs, int64_t m, double *ys) {
    TGP_ARG(k && X && alpha && Xs && ys && n > 0 && m > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t io_bytes = rup(2 * n * 8) + rup(n * 8) + rup(2 * m * 8) + rup(m * 8);
    int rc = ensure_io(ctx, io_bytes);
    if (rc) return rc;
    rc = ensure_hio(ctx, io_bytes);
    if (rc) return rc;
    Arena ar{(char *)ext_of(ctx)->io.buf};
    double *d_X = ar.take<double>(2 * n), *d_a = ar.take<double>(n), *d_Xs = ar.take<double>(2 * m),
           *d_ys = ar.take<double>(m);
    TGP_HIP(h2d(ctx, d_X, X, 2 * n * 8));
    TGP_HIP(h2d(ctx, d_a, alpha, n * 8));
    TGP_HIP(h2d(ctx, d_Xs, Xs, 2 * m * 8));
    rc = tgp_d_gp_predict(ctx, k, d_X, n, d_a, d_Xs, m, d_ys);
    if (rc) return rc;
    TGP_HIP(d2h_sync(ctx, ys, d_ys, m * 8));
    return 0;
}

// ---- S1 ---------------------------------------------------------------------------------------
int tgp_kernel_matrix(tgp_ctx *ctx, const tgp_kernel *k, const double *X, int64_t n, const double *Y, int64_t m,
                      double *out) {
    TGP_ARG(k && X && out && n > 0);
    const int self = (Y == nullptr);
    if (self) m = n;
    TGP_ARG(m > 0);
    TGP_HIP(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    int rc = ensure_io(ctx, rup(2 * n * 8) + rup(2 * m * 8) + rup((size_t)n * m * 8));
    if (rc) return rc;
    Arena ar{(char *)ext_of(ctx)->io.buf};
    double *d_X = ar.take<double>(2 * n), *d_Y = ar.take<double>(2 * m), *d_o = ar.take<double>((size_t)n * m);
    TGP_HIP(hipMemcpyAsync(d_X, X, 2 * n * 8, hipMemcpyHostToDevice, st));
    if (!self) TGP_HIP(hipMemcpyAsync(d_Y, Y, 2 * m * 8, hipMemcpyHostToDevice, st));
    rc = launch_kernel_dense(ctx, k, d_X, n, self ? d_X : d_Y, m, self, d_o);
    if (rc) return rc;
    TGP_HIP(hipMemcpyAsync(out, d_o, (size_t)n * m * 8, hipMemcpyDeviceToHost, st));
    TGP_HIP(hipStreamSynchronize(st));
    return 0;
}

}  // extern "C"
