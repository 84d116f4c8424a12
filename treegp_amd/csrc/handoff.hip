// Cross-stream hand-offs of the factorisation's schedules (chol.hip, dist.hip).
// A one-thread kernel queued behind the producer on its stream publishes a sequence number; the consumer stream waits for
// it with hipStreamWaitValue32.  From the producer's last kernel ending to the consumer's first kernel starting: 2.7 us,
// against 10 us for hipEventRecord + hipStreamWaitEvent (and 4.4 us for hipStreamWriteValue32; tools/probes/
// waitvalue_probe.hip, profiles/r03_waitvalue_probe.txt).  The panel chain of the chain-bound sizes crosses streams twice
// per pair of panels: Cholesky -9 % at N = 3000, -5 % at 8192.
//
// A stream parked in a wait-value is only safe where another stream's work can still be dispatched: a tool that
// serialises dispatches across queues (rocprofv3 --pmc does) never runs the signalling kernel, and the process hangs.
// So the mechanism is tried once per device before it is used (waitvalue_selftest): the consumer parks FIRST on a flag in
// pinned host memory, the producer's kernel is queued second, and if the wait has not been released after 2 s the host
// writes the flag itself and every context of the process hands over by events instead.  TGP_SYNC_EVENTS=1 forces events
// (the documented profiler setting), TGP_SYNC_EVENTS=0 forces flags without the trial.
#include "tgp_internal.h"

#include <chrono>
#include <mutex>
#include <thread>

namespace {
__global__ void signal_kernel(unsigned *flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool waitvalue_selftest(int device) {
    int can = 0;
    if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !can) return false;
    unsigned *flag = nullptr;
    if (hipHostMalloc((void **)&flag, 64, hipHostMallocDefault) != hipSuccess) return false;
    __atomic_store_n(flag, 0u, __ATOMIC_SEQ_CST);
    hipStream_t a = nullptr, b = nullptr;
    bool ok = false;
    if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&b, hipStreamNonBlocking) == hipSuccess) {
        if (hipStreamWaitValue32(b, flag, 1u, hipStreamWaitValueGte, 0xffffffffu) == hipSuccess) {
            signal_kernel<<<1, 1, 0, b>>>(flag + 8, 1u);         // a kernel queued BEHIND the parked wait, as in real use
            signal_kernel<<<1, 1, 0, a>>>(flag, 1u);
            if (hipGetLastError() == hipSuccess) {
                const auto t0 = std::chrono::steady_clock::now();
                while (!(ok = hipStreamQuery(b) == hipSuccess) &&
                       std::chrono::steady_clock::now() - t0 < std::chrono::seconds(2))
                    std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            if (!ok) __atomic_store_n(flag, 1u, __ATOMIC_SEQ_CST);      // release the parked wait from the host
            (void)hipStreamSynchronize(b);
            (void)hipStreamSynchronize(a);
        } else {
            (void)hipGetLastError();
        }
    }
    if (a) (void)hipStreamDestroy(a);
    if (b) (void)hipStreamDestroy(b);
    (void)hipHostFree(flag);
    return ok;
}

bool device_hands_over_by_flags(int device) {
    static std::mutex mu;
    static int verdict[64] = {0};                  // 0 untried, 1 flags, 2 events
    std::lock_guard<std::mutex> lock(mu);
    int &v = verdict[device & 63];
    if (v == 0) {
        const char *e = getenv("TGP_SYNC_EVENTS");
        // rocprofv3's counter collection exports ROCPROF_COUNTER_COLLECTION and serialises dispatches: the one known case in
        // which a parked wait never sees its signalling kernel.  The trial below has never been run under it without
        // TGP_SYNC_EVENTS=1 (every --pmc script sets it) and uses two fresh streams, whose queue mapping need not be that of
        // the real bulk and chain streams: the variable stays an events trigger of its own.
        const char *pmc = getenv("ROCPROF_COUNTER_COLLECTION");
        if (e && *e) v = atoi(e) != 0 ? 2 : 1;
        else if (pmc && *pmc && atoi(pmc) != 0) v = 2;
        else v = waitvalue_selftest(device) ? 1 : 2;
    }
    return v == 1;
}
}  // namespace

bool tgp_handoff_by_flags(tgp_ctx *ctx) {
    if (ctx->handoff == 0) ctx->handoff = device_hands_over_by_flags(ctx->device) ? 1 : 2;
    return ctx->handoff == 1;
}

// the next sequence number of flag `id` of this context (monotonic over the context's life)
unsigned tgp_next_seq(tgp_ctx *ctx, int id, hipError_t *err) {
    *err = hipSuccess;
    if (ctx->flag_seq[id] >= 0xfffffff0u) {
        // the sequence numbers are about to wrap (days of continuous use): let everything queued finish -- every wait that
        // was issued is then satisfied -- and start all flags of this context again from zero
        hipError_t e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemset(ctx->d_flags, 0, 16 * 64);
        if (e != hipSuccess) {
            *err = e;
            return 0;
        }
        for (unsigned &q : ctx->flag_seq) q = 0;
        ctx->head_count = 0;
    }
    return ++ctx->flag_seq[id];
}

hipError_t tgp_signal_value(tgp_ctx *ctx, hipStream_t from, int id, unsigned v) {
    signal_kernel<<<1, 1, 0, from>>>(ctx->d_flags + 16 * id, v);
    return hipGetLastError();
}

// after everything queued on `from` so far
hipError_t tgp_signal(tgp_ctx *ctx, hipStream_t from, int id, hipEvent_t ev) {
    if (!tgp_handoff_by_flags(ctx)) return hipEventRecord(ev, from);
    hipError_t e = hipSuccess;
    const unsigned v = tgp_next_seq(ctx, id, &e);
    return e != hipSuccess ? e : tgp_signal_value(ctx, from, id, v);
}

// `to` proceeds once the last tgp_signal on this id has happened
hipError_t tgp_await(tgp_ctx *ctx, hipStream_t to, int id, hipEvent_t ev) {
    if (!tgp_handoff_by_flags(ctx)) return hipStreamWaitEvent(to, ev, 0);
    return hipStreamWaitValue32(to, ctx->d_flags + 16 * id, ctx->flag_seq[id], hipStreamWaitValueGte, 0xffffffffu);
}

extern "C" int tgp_handoff_mode(tgp_ctx *ctx) {
    if (!ctx) return -1;
    if (hipSetDevice(ctx->device) != hipSuccess) return -2;
    return tgp_handoff_by_flags(ctx) ? 1 : 2;
}
