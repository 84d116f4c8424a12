// Do fp64 MFMA and fp64 VALU FMAs overlap on gfx950?  (Both are rated 78.6 TFLOP/s; if they are separate pipes, vector FMAs
// issued in the shadow of a 64-clock v_mfma_f64_16x16x4 would be free.)  Per iteration a wave issues NM MFMAs (independent
// accumulators) and NV v_fma_f64 (independent chains); 256 threads x 1024 workgroups, 2 workgroups per CU resident... reports
// time per iteration in clocks at the in-kernel clock, and the combined TFLOP/s.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NM, int NV>
__global__ __launch_bounds__(256, 2) void k(double *out, unsigned long long *stamps, int iters, double seed) {
    const int lane = threadIdx.x & 63;
    d4 acc[NM > 0 ? NM : 1];
    constexpr int NC = NV > 32 ? 32 : (NV > 0 ? NV : 1);          // independent VALU chains, used round-robin
    double v[NC];
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = (d4){0, 0, 0, 0};
    for (int i = 0; i < NC; ++i) v[i] = seed + i + lane * 1e-3;
    const double a = 1.0 + lane * 1e-3, b = 0.5 - lane * 1e-3, c = 1.0 - 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int PER = NM > 0 ? (NV + NM - 1) / NM : NV;       // VALU FMAs behind each MFMA
#pragma unroll
        for (int m = 0; m < (NM > 0 ? NM : 1); ++m) {
            if (NM > 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int i = m * PER + q;
                if (i < NV) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[i % NC]) : "v"(c), "v"(b));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][3];
    for (int i = 0; i < NC; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NM, int NV>
void run(double *out, unsigned long long *st) {
    const int blocks = 1024, iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NM, NV><<<blocks, 256>>>(out, st, iters, 1.0);
    (void)hipEventRecord(e0);
    for (int l = 0; l < 5; ++l) k<NM, NV><<<blocks, 256>>>(out, st, iters, 1.0);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[2];
    (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double clk = (double)h[0] / (double)h[1] * 100e6;
    const double waves = 5.0 * blocks * 4, mf = waves * iters * NM * 2048.0, vf = waves * iters * NV * 128.0;
    printf("NM %2d MFMA + NV %3d v_fma_f64 per iteration: %7.2f ms  MFMA %.1f TF + VALU %.1f TF = %.1f TF   clock %.2f GHz   %.0f clk per iteration per wave (2 waves per SIMD)\n",
           NM, NV, ms, mf / ms / 1e9, vf / ms / 1e9, (mf + vf) / ms / 1e9, clk / 1e9, (double)h[0] / iters);
}

int main() {
    double *out; unsigned long long *st;
    (void)hipMalloc(&out, 1024 * 256 * 8); (void)hipMalloc(&st, 1024 * 16);
    run<16, 0>(out, st);
    run<0, 64>(out, st);
    run<16, 16>(out, st);
    run<16, 64>(out, st);
    run<16, 128>(out, st);
    run<16, 256>(out, st);
    run<8, 128>(out, st);
    run<16, 0>(out, st);
    return 0;
}
