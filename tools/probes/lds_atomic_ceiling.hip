// the pair histogram is "VALU + LDS-atomic bound; report pairs/s vs a measured LDS-atomic ceiling".  This is that
// ceiling: per-wave private fp64 histograms in LDS (as in csrc/kk.hip), every lane issuing ds_add_f64 back to back with
// nothing else to do, for the bin distributions the two pair kernels produce:
//   log20    20 log-spaced bins in [1/sqrt(N), 0.5] of uniform points: bin populations grow like r^2, the last 3 bins take
//            ~70 % of the pairs (heavy same-address serialisation inside a wave)
//   twod441  21 x 21 pixels, uniform over the pixels (pairs inside the box are nearly uniform in dx, dy)
//   spread   every lane its own address (no conflicts): the instruction-rate limit
// NACC accumulators per pair are added to NACC copies of the histogram (5 for log bins, 3 for TwoD), as the kernels do.
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_atomic_ceiling tools/probes/lds_atomic_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

template <int NACC>
__global__ __launch_bounds__(256) void hist(const int *__restrict__ bins, int nb, int iters, double *out) {
    extern __shared__ double h[];                         // [4 waves][NACC][nb]
    for (int i = threadIdx.x; i < 4 * NACC * nb; i += 256) h[i] = 0.0;
    __syncthreads();
    double *my = h + (threadIdx.x >> 6) * NACC * nb;
    const int *b = bins + (size_t)(blockIdx.x * 256 + threadIdx.x) * iters;
    for (int it = 0; it < iters; ++it) {
        const int k = b[it];
#pragma unroll
        for (int a = 0; a < NACC; ++a) __hip_atomic_fetch_add(my + a * nb + k, 1.0 + a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = h[0] + h[nb];
}

int main() {
    const int wgs = 256 * 8, iters = 2048;
    const size_t n = (size_t)wgs * 256 * iters;
    std::vector<int> hb(n);
    int *d_b; double *d_o;
    hipMalloc(&d_b, n * 4); hipMalloc(&d_o, wgs * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    struct Case { const char *name; int nb, nacc; int kind; } cases[] = {{"log20", 20, 5, 0}, {"twod441", 441, 3, 1}, {"spread", 64, 1, 2}};
    for (const Case &c : cases) {
        srand(7);
        for (size_t i = 0; i < n; ++i) {
            if (c.kind == 0) {            // r uniform in area up to 0.5: P(r < x) ~ x^2; log bins from 1/181 to 0.5
                const double u = (double)rand() / RAND_MAX, r = 0.5 * sqrt(u), lo = 1.0 / 181.0;
                int k = r <= lo ? 0 : (int)(log(r / lo) / (log(0.5 / lo) / 20));
                hb[i] = k < 0 ? 0 : (k > 19 ? 19 : k);
            } else if (c.kind == 1) hb[i] = rand() % 441;
            else hb[i] = (int)(i % 64);   // lane l of a wave always bin l
        }
        hipMemcpy(d_b, hb.data(), n * 4, hipMemcpyHostToDevice);
        const size_t lds = (size_t)4 * c.nacc * c.nb * 8;
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            if (c.nacc == 5) hist<5><<<wgs, 256, lds>>>(d_b, c.nb, iters, d_o);
            else if (c.nacc == 3) hist<3><<<wgs, 256, lds>>>(d_b, c.nb, iters, d_o);
            else hist<1><<<wgs, 256, lds>>>(d_b, c.nb, iters, d_o);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        printf("%-8s %3d bins x %d accumulators: %.3e atomics/s = %.3e pairs/s  (%.2f ms)\n", c.name, c.nb, c.nacc,
               (double)n * c.nacc / (best * 1e-3), (double)n / (best * 1e-3), best);
    }
    return 0;
}
