// Single-wave issue cost and dependent latency of the instructions potrf128's Gauss-Jordan sweep is made of (one wave per SIMD,
// nothing else on the CU -- the sweep's situation): s_memtime ticks per instruction for chains of DEPENDENT instructions and for
// INDEPENDENT ones (8 registers in rotation).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/probes/issue_probe tools/probes/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int WHICH>
__global__ __launch_bounds__(64) void probe(double *out, unsigned long long *cyc, int iters) {
    double a0 = 1.0 + threadIdx.x * 1e-9, a1 = a0 + 1e-9, a2 = a0 + 2e-9, a3 = a0 + 3e-9, a4 = a0 + 4e-9, a5 = a0 + 5e-9, a6 = a0 + 6e-9, a7 = a0 + 7e-9;
    double b = 1.0000001, c = 1e-12;
    int x = threadIdx.x, yv = 3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (WHICH == 0) {          // dependent v_fma_f64
            asm volatile(REP64("v_fma_f64 %0, %0, %1, %2\n\t") : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (WHICH == 1) {   // independent v_fma_f64
            asm volatile(REP8("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                              "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (WHICH == 2) {   // dependent v_fmac_f64_dpp (accumulator chain; DPP source constant)
            asm volatile(REP64("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a0) : "v"(b), "v"(c));
        } else if constexpr (WHICH == 3) {   // independent v_fmac_f64_dpp
            asm volatile(REP8("v_fmac_f64_dpp %0, %8, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %8, %9 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %2, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %4, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %5, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                              "v_fmac_f64_dpp %6, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %7, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        } else if constexpr (WHICH == 4) {   // dependent v_rsq_f64
            asm volatile(REP64("v_rsq_f64 %0, %0\n\t") : "+v"(a0));
        } else if constexpr (WHICH == 5) {   // independent v_rsq_f64
            asm volatile(REP8("v_rsq_f64 %0, %0\n\tv_rsq_f64 %1, %1\n\tv_rsq_f64 %2, %2\n\tv_rsq_f64 %3, %3\n\tv_rsq_f64 %4, %4\n\tv_rsq_f64 %5, %5\n\tv_rsq_f64 %6, %6\n\tv_rsq_f64 %7, %7\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if constexpr (WHICH == 6) {   // dependent v_cndmask_b32 (vcc constant)
            asm volatile(REP64("v_cndmask_b32 %0, %0, %1, vcc\n\t") : "+v"(x) : "v"(yv) : "vcc");
        } else if constexpr (WHICH == 7) {   // dependent v_mov_b32_dpp
            asm volatile(REP64("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t") : "+v"(x));
        } else if constexpr (WHICH == 8) {   // dependent v_mul_f64
            asm volatile(REP64("v_mul_f64 %0, %0, %1\n\t") : "+v"(a0) : "v"(b));
        } else if constexpr (WHICH == 9) {   // v_rsq_f64 followed by a dependent v_mul_f64 (the pivot chain's first link)
            asm volatile(REP64("v_rsq_f64 %0, %0\n\tv_mul_f64 %0, %0, %1\n\t") : "+v"(a0) : "v"(b));
        } else if constexpr (WHICH == 10) {  // dependent fmac_dpp whose DPP source is the previous result (s_nop 1 between, as the sweep needs)
            asm volatile(REP64("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t") : "+v"(a0) : "v"(c));
        } else if constexpr (WHICH == 11) {  // independent v_mov_b64
            asm volatile(REP8("v_mov_b64 %0, %8\n\tv_mov_b64 %1, %8\n\tv_mov_b64 %2, %8\n\tv_mov_b64 %3, %8\n\tv_mov_b64 %4, %8\n\tv_mov_b64 %5, %8\n\tv_mov_b64 %6, %8\n\tv_mov_b64 %7, %8\n\t")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else if constexpr (WHICH == 12) {  // v_mfma_f64_16x16x4 dependent chain
            typedef double d4 __attribute__((ext_vector_type(4)));
            d4 acc = {a0, a1, a2, a3};
            for (int k = 0; k < 64; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, acc, 0, 0, 0);
            a0 = acc[0] + acc[1]; a1 = acc[2]; a2 = acc[3];
        } else if constexpr (WHICH == 13) {  // v_mfma_f64_4x4x4 dependent chain
            for (int k = 0; k < 64; ++k) a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(b, c, a0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <int W>
void run(const char *what, double *d_out, unsigned long long *d_cyc) {
    const int iters = 200;
    probe<W><<<1, 64>>>(d_out, d_cyc, iters);
    probe<W><<<1, 64>>>(d_out, d_cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c = 0;
    hipMemcpy(&c, d_cyc, 8, hipMemcpyDeviceToHost);
    printf("%-58s %7.2f ticks per instruction (pair for 9)\n", what, (double)c / (iters * 64.0));
}

int main() {
    double *d_out; unsigned long long *d_cyc;
    hipMalloc(&d_out, 64 * 8); hipMalloc(&d_cyc, 8);
    run<0>("v_fma_f64 dependent", d_out, d_cyc);
    run<1>("v_fma_f64 independent (8 in rotation)", d_out, d_cyc);
    run<8>("v_mul_f64 dependent", d_out, d_cyc);
    run<2>("v_fmac_f64_dpp dependent accumulator", d_out, d_cyc);
    run<3>("v_fmac_f64_dpp independent (8 in rotation)", d_out, d_cyc);
    run<10>("s_nop 1 + v_fmac_f64_dpp, DPP source = previous result", d_out, d_cyc);
    run<4>("v_rsq_f64 dependent", d_out, d_cyc);
    run<5>("v_rsq_f64 independent", d_out, d_cyc);
    run<9>("v_rsq_f64 + dependent v_mul_f64", d_out, d_cyc);
    run<6>("v_cndmask_b32 dependent", d_out, d_cyc);
    run<7>("s_nop 1 + v_mov_b32_dpp dependent", d_out, d_cyc);
    run<11>("v_mov_b64 independent", d_out, d_cyc);
    run<12>("v_mfma_f64_16x16x4_f64 dependent", d_out, d_cyc);
    run<13>("v_mfma_f64_4x4x4_4b_f64 dependent", d_out, d_cyc);
    return 0;
}
