// the von Karman K build is fp64-VALU-bound; this is the register-only ceiling it is normalised by.
// Every thread evaluates vonkarman_unit (the library's own evaluator, csrc/bessel_k56.h) for ELEMS arguments drawn like the
// separations of configs[2] (uniform points in the unit square, length scale 0.1: u = d / 0.1) and adds the results up in a
// register; nothing is stored but one double per thread.  Variants: Chebyshev table gathered from global memory / from an
// LDS copy; arguments all in the Chebyshev range (u > 0.16) or as drawn.
// build: hipcc --offload-arch=gfx950 -O3 -I treegp_amd/csrc -o /tmp/vk_ceiling tools/probes/vk_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "bessel_k56.h"

template <bool LDS>
__global__ __launch_bounds__(256) void vk_eval(const double *__restrict__ u0, int elems, double *out) {
    __shared__ double tab[6 * K56_NDEG];
    if (LDS) {
        vonkarman_stage_table(tab);
        __syncthreads();
    }
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    double u = u0[tid], acc = 0.0;
    for (int e = 0; e < elems; ++e) {
        acc += LDS ? vonkarman_unit_tab(u, tab) : vonkarman_unit(u);
        u = u * 1.0000001 + 1e-7 * (double)(e & 7);          // keeps the argument (and its octave) lane-dependent and moving
    }
    out[tid] = acc;
}

int main() {
    const int threads = 256 * 256 * 16, elems = 2048;
    double *h = (double *)malloc(threads * 8), *d_u, *d_o;
    srand(1);
    for (int i = 0; i < threads; ++i) {
        const double dx = (double)rand() / RAND_MAX - (double)rand() / RAND_MAX, dy = (double)rand() / RAND_MAX - (double)rand() / RAND_MAX;
        h[i] = sqrt(dx * dx + dy * dy) / 0.1;
    }
    hipMalloc(&d_u, threads * 8); hipMalloc(&d_o, threads * 8);
    hipMemcpy(d_u, h, threads * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int v = 0; v < 2; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(a);
            if (v) vk_eval<true><<<threads / 256, 256>>>(d_u, elems, d_o); else vk_eval<false><<<threads / 256, 256>>>(d_u, elems, d_o);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (ms < best) best = ms;
        }
        printf("register-only von Karman evaluation, table in %s: %.3e elements/s (%.2f ms for %.3e elements)\n", v ? "LDS" : "global memory",
               (double)threads * elems / (best * 1e-3), best, (double)threads * elems);
    }
    return 0;
}
