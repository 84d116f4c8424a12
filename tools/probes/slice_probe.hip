// Timing and bits of the two latency tiles (gemm_tile.h: nt_small_tile, operands straight from global memory in MFMA layout;
// nt_slice_tile, staged through LDS with whole-row loads) on what the panel chain runs: eight workgroups, one 16 x 128 slice
// each of X = A W^T (K = 128) or C -= P P^T (K = 256), operands cold (a fresh region of a large buffer per repetition).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I treegp_amd/csrc -o tools/probes/slice_probe tools/probes/slice_probe.hip
#include "gemm_tile.h"
#include <cstdio>
#include <vector>
template <int WHICH, int MODE, int K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void probe(const double *A, const double *B, double *C, int ldb, unsigned long long *ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int64_t o = (int64_t)blockIdx.x * 16 * 256;
    if (WHICH == 0) nt_small_tile<MODE, K, 1>(A + o, 256, B, ldb, C + o, 256, nullptr, nullptr);
    else nt_slice_tile<MODE, K, 1>(nt_slice_lds_storage(), A + o, 256, B, ldb, C + o, 256, nullptr, nullptr);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) ticks[blockIdx.x] = __builtin_amdgcn_s_memrealtime() - t0;
}
template <int WHICH, int MODE, int K>
double run(const double *dA, const double *dB, double *dC, int ldb, unsigned long long *dT, int reps, size_t region, std::vector<double> *out) {
    double sum = 0;
    for (int r = 0; r < reps; ++r) {
        const size_t off = (size_t)r * region;
        probe<WHICH, MODE, K><<<8, 256>>>(dA + off, dB + off, dC + off, ldb, dT);
        unsigned long long h[8];
        hipMemcpy(h, dT, sizeof(h), hipMemcpyDeviceToHost);
        unsigned long long m = 0;
        for (auto v : h) m = v > m ? v : m;
        if (r) sum += (double)m / 100.0;
    }
    if (out) {
        out->resize(128 * 256);
        hipMemcpy(out->data(), dC, out->size() * 8, hipMemcpyDeviceToHost);
    }
    return sum / (reps - 1);
}
int main() {
    const int reps = 40;
    const size_t region = 1 << 20;                 // doubles between the repetitions' operands (8 MB: nothing of the last one is cached)
    const size_t n = region * reps + (1 << 17);
    std::vector<double> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5;
    double *dA, *dB, *dC0, *dC1;
    unsigned long long *dT;
    hipMalloc(&dA, n * 8); hipMalloc(&dB, n * 8); hipMalloc(&dC0, n * 8); hipMalloc(&dC1, n * 8); hipMalloc(&dT, 64);
    hipMemcpy(dA, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dB, h.data() + 7, (n - 7) * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC0, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC1, h.data(), n * 8, hipMemcpyHostToDevice);
    std::vector<double> c0, c1;
    const double a = run<0, 0, 128>(dA, dB, dC0, 128, dT, reps, region, &c0);
    const double b = run<1, 0, 128>(dA, dB, dC1, 128, dT, reps, region, &c1);
    size_t bad = 0;
    for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
    printf("X = A W^T, K = 128 (ldb 128): direct %.2f us, staged %.2f us per slice (slowest of 8 workgroups, mean of %d cold runs); %zu of %zu values differ\n", a, b, reps - 1, bad, c0.size());
    const double c = run<0, 1, 256>(dA, dB, dC0, 256, dT, reps, region, &c0);
    const double d = run<1, 1, 256>(dA, dB, dC1, 256, dT, reps, region, &c1);
    bad = 0;
    for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
    printf("C -= P P^T, K = 256 (ldb 256): direct %.2f us, staged %.2f us per slice; %zu of %zu values differ\n", c, d, bad, c0.size());
    const double e = run<0, 1, 128>(dA, dB, dC0, 256, dT, reps, region, &c0);
    const double f = run<1, 1, 128>(dA, dB, dC1, 256, dT, reps, region, &c1);
    bad = 0;
    for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
    printf("C -= X X^T, K = 128 (ldb 256): direct %.2f us, staged %.2f us per slice; %zu of %zu values differ\n", e, f, bad, c0.size());
    return 0;
}
