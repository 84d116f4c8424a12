// Issue rate of the fp64 VALU instructions the fused predict's 2^t is made of, against v_fma_f64: 256 threads x 2048 workgroups,
// 8 independent chains per thread, 4096 iterations.  Prints wave-instructions per clock per SIMD (1/4 = full rate for wave64).
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate_probe tools/probes/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAINS 8
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void probe(double *out, double seed, int shift) {
    double v[CHAINS];
    int iv[CHAINS];
    for (int c = 0; c < CHAINS; ++c) { v[c] = seed + c * 0.37 + threadIdx.x * 1e-3; iv[c] = threadIdx.x + c; }
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v[c]) : "v"(seed));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed));
            if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed));
            if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[c]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(iv[c]) : "v"(v[c]));
            if (OP == 5) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[c]) : "v"(shift));
            if (OP == 6) asm volatile("v_lshl_add_u32 %0, %0, 12, %1" : "+v"(iv[c]) : "v"(shift));
            if (OP == 7) asm volatile("v_and_b32 %0, %0, %1" : "+v"(iv[c]) : "v"(shift));
            if (OP == 8) asm volatile("v_min_f64 %0, %0, %1" : "+v"(v[c]) : "v"(seed));
            if (OP == 9) asm volatile("v_fract_f64 %0, %0" : "+v"(v[c]));
            if (OP == 10) asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(iv[c]) : "v"(shift));
            if (OP == 11) asm volatile("v_ashrrev_i32 %0, 8, %0" : "+v"(iv[c]));
        }
    }
    double s = 0.0;
    for (int c = 0; c < CHAINS; ++c) s += v[c] + iv[c];
    if (s == 1.2345e-300) out[0] = s;
}

template <int OP>
void run(const char *name, double *out, double clock_hz) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    probe<OP><<<2048, 256>>>(out, 1.0000001, 0);
    hipEventRecord(a);
    probe<OP><<<2048, 256>>>(out, 1.0000001, 0);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double wave_instr = 2048.0 * 4 * CHAINS * ITERS;                 // per launch
    const double per_simd = wave_instr / (256.0 * 4);                       // 256 CUs x 4 SIMDs
    printf("%-22s %8.3f ms  %.3f wave-instr/clk/SIMD at %.2f GHz nominal (clk per wave-instr %.2f)\n", name, ms,
           per_simd / (ms * 1e-3 * clock_hz), clock_hz * 1e-9, ms * 1e-3 * clock_hz / per_simd);
}

int main() {
    double *out;
    hipMalloc(&out, 64);
    const double f = 2.4e9;
    run<0>("v_fma_f64", out, f);
    run<1>("v_add_f64", out, f);
    run<2>("v_mul_f64", out, f);
    run<8>("v_min_f64", out, f);
    run<3>("v_rndne_f64", out, f);
    run<9>("v_fract_f64", out, f);
    run<4>("v_cvt_i32_f64", out, f);
    run<5>("v_ldexp_f64", out, f);
    run<6>("v_lshl_add_u32", out, f);
    run<7>("v_and_b32", out, f);
    run<10>("v_lshlrev_b32_sdwa", out, f);
    run<11>("v_ashrrev_i32", out, f);
    return 0;
}
