// Microbenchmark: practical fp64 MFMA ceiling of the chip under DVFS (development aid).
//   mode 0: registers only, 16 independent accumulators per wave
//   mode 1: + the syrk tile's LDS fragment reads (8 ds_read per 16 MFMAs)
// Reports TFLOP/s and the in-kernel clock (s_memtime / s_memrealtime, 100 MHz reference).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int RANDOM>
__global__ __launch_bounds__(256, 2) void peak_kernel(double *out, unsigned long long *stamps, int iters) {
    __shared__ double lds[2 * 128 * 17];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    for (int i = tid; i < 2 * 128 * 17; i += 256) { unsigned h = (i + 7919u * blockIdx.x) * 2654435761u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16; lds[i] = RANDOM ? ((double)h / 2147483648.0 - 1.0) * (1.0 + 1e-9 * (h & 1023)) : 1e-3 * (i % 97); }
    __syncthreads();
    d4 acc[4][4];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = (d4){0, 0, 0, 0};
    double af[4], bf[4];
    for (int m = 0; m < 4; ++m) { af[m] = RANDOM ? lds[(lane * 37 + m * 501) % 4000] : 1.0 + lane * 1e-3 + m; bf[m] = RANDOM ? lds[(lane * 53 + m * 777 + 11) % 4000] : 0.5 - lane * 1e-3 + m; }
    const int fa = ((w >> 1) * 64 + (lane & 15)) * 17 + (lane >> 4);
    const int fb = 128 * 17 + ((w & 1) * 64 + (lane & 15)) * 17 + (lane >> 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            if (MODE == 1) {
#pragma unroll
                for (int m = 0; m < 4; ++m) af[m] = lds[fa + m * 16 * 17 + k4 * 4];
#pragma unroll
                for (int n = 0; n < 4; ++n) bf[n] = lds[fb + n * 16 * 17 + k4 * 4];
            }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[m], bf[n], acc[m][n], 0, 0, 0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
    const int blocks = 512, iters = 4000;
    double *out; unsigned long long *st;
    hipMalloc(&out, blocks * 256 * 8); hipMalloc(&st, blocks * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0);
            for (int l = 0; l < 10; ++l) {
                if (mode == 0) peak_kernel<0, 0><<<blocks, 256>>>(out, st, iters);
                else if (mode == 1) peak_kernel<1, 0><<<blocks, 256>>>(out, st, iters);
                else if (mode == 2) peak_kernel<0, 1><<<blocks, 256>>>(out, st, iters);
                else peak_kernel<1, 1><<<blocks, 256>>>(out, st, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(blocks * 2);
            hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
            double clk = 0; for (int b = 0; b < blocks; ++b) clk += (double)h[2 * b] / (double)h[2 * b + 1] * 100e6; clk /= blocks;
            double flops = 10.0 * blocks * 4 * iters * 64.0 * 2048.0;
            if (rep >= 3) printf("mode %d (2,3 = random operands): %.2f ms  %.2f TFLOP/s  in-kernel clock %.3f GHz  -> %.1f cycles per MFMA per SIMD\n", mode, ms,
                   flops / (ms * 1e-3) / 1e12, clk / 1e9, (double)h[0] / (iters * 64.0 * 2));
        }
    }
    return 0;
}
