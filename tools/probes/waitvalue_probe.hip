// Does hipStreamWaitValue32 work here, and how soon after a RUNNING kernel writes the flag does a kernel on another stream
// start?  Kernel A (one workgroup per CU, ~2 ms of spinning) writes the flag (release, system scope) after ~1 ms; stream 2 waits
// for the value and then launches kernel B, which stamps the time.  Also the same with an event recorded after A (the baseline).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/waitvalue_probe tools/probes/waitvalue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void kernel_a(unsigned *flag, unsigned long long *out, long long half_ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t0;
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < half_ticks) {}
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        out[1] = __builtin_amdgcn_s_memrealtime();
        __threadfence_system();
        __hip_atomic_store(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < 2 * half_ticks) {}
    if (blockIdx.x == 0 && threadIdx.x == 0) out[2] = __builtin_amdgcn_s_memrealtime();
}
__global__ void signal_kernel(unsigned *flag, unsigned v) {
    __hip_atomic_store(flag, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void kernel_b(unsigned long long *out) {
    if (threadIdx.x == 0) out[3] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    unsigned *flag = nullptr;
    unsigned long long *out, h[4];
    hipError_t e = hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal memory): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) { hipMalloc((void **)&flag, 64); }
    hipMalloc((void **)&out, 64);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(flag, 0, 64);
        hipMemset(out, 0, 64);
        hipDeviceSynchronize();
        if (mode == 0) {
            e = hipStreamWaitValue32(s2, flag, 1u, hipStreamWaitValueGte, 0xffffffffu);
            printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
            if (e != hipSuccess) continue;
            kernel_b<<<1, 64, 0, s2>>>(out);
            kernel_a<<<128, 64, 0, s1>>>(flag, out, 100000);          // 1 ms + 1 ms (s_memrealtime: 100 MHz)
        } else if (mode == 2) {          // flag written by the stream itself after the kernel (hipStreamWriteValue32)
            hipStreamWaitValue32(s2, flag, 2u, hipStreamWaitValueGte, 0xffffffffu);
            kernel_b<<<1, 64, 0, s2>>>(out);
            kernel_a<<<128, 64, 0, s1>>>(flag, out, 100000);
            e = hipStreamWriteValue32(s1, flag, 2u, 0);
            printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e));
        } else if (mode == 3) {          // flag written by a one-thread kernel queued behind the producer
            hipStreamWaitValue32(s2, flag, 3u, hipStreamWaitValueGte, 0xffffffffu);
            kernel_b<<<1, 64, 0, s2>>>(out);
            kernel_a<<<128, 64, 0, s1>>>(flag, out, 100000);
            signal_kernel<<<1, 1, 0, s1>>>(flag, 3u);
        } else {
            hipEvent_t ev;
            hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            kernel_a<<<128, 64, 0, s1>>>(flag, out, 100000);
            hipEventRecord(ev, s1);
            hipStreamWaitEvent(s2, ev, 0);
            kernel_b<<<1, 64, 0, s2>>>(out);
        }
        e = hipDeviceSynchronize();
        hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
        printf("%s: %s; A start 0, flag written at %.1f us, A end %.1f us, B started at %.1f us\n",
               mode == 0 ? "stream wait-value on a flag written mid-kernel" : mode == 1 ? "event after the kernel" :
               mode == 2 ? "stream write-value after the kernel + wait-value" : "one-thread signal kernel after the kernel + wait-value",
               hipGetErrorString(e),
               (h[1] - h[0]) * 0.01, (h[2] - h[0]) * 0.01, ((long long)h[3] - (long long)h[0]) * 0.01);
    }
    return 0;
}
