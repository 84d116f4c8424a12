// von Karman covariance profile  f(u) = u^(5/6) K_{5/6}(2 pi u) / lim0,  f(0) = 1  in fp64
// (treegp/kernels.py:253-262; lim0 = Gamma(5/6) / (2 pi^(5/6))).
//
// SciPy evaluates K_nu with AMOS zbesk; that source is not part of the reference, so the
// function is restated from the published formulas for the single order nu = 5/6:
//   x = 2 pi u <= 1 : ascending series (A&S 9.6.2 / 9.6.10).  With t = pi^2 u^2,
//                     f = S1(t) - D u^(5/3) S2(t)  -- the u^(-5/6) of I_{-nu} cancels the
//                     prefactor, so no pow() is needed, only cbrt.
//   x > 1           : g(x) = e^x sqrt(x) K_nu(x) by piecewise Chebyshev (Clenshaw), tables
//                     from gen_bessel_table.py;  f = PRE * cbrt(u) * exp(-x) * g(x).
//   x > 697.87388   : 0, the argument above which SciPy/AMOS reports underflow (probed).
// Host-compilable (tests build it with g++ and compare against scipy.special.kv).
#pragma once
#include <cmath>
#ifndef __HIPCC__
#define __device__
#define __host__
#define __forceinline__ inline
#endif
#include "bessel_k56_coeffs.h"

#define K56_XMAX 697.8738840444552

// `cheb`: where the 6 x K56_NDEG Chebyshev table is read from.  The octave index differs from lane to lane, so on the device
// every coefficient fetch is a gather; from global memory that is 24 gathers per element and costs more than the ~120
// fp64 operations around them -- kernels that evaluate many elements per workgroup stage the table (1.2 KB) in LDS with
// vonkarman_stage_table and pass that copy.
__host__ __device__ __forceinline__ double vonkarman_unit_tab(double u, const double *cheb) {
    if (u == 0.0) return 1.0;
    const double x = K56_TWO_PI * u;
    if (x <= 1.0) {
        const double t = K56_PI2 * u * u;
        double s1 = k56_s1[K56_NSER - 1], s2 = k56_s2[K56_NSER - 1];
#pragma unroll
        for (int k = K56_NSER - 2; k >= 0; --k) {
            s1 = fma(s1, t, k56_s1[k]);
            s2 = fma(s2, t, k56_s2[k]);
        }
        const double cr = cbrt(u);
        return s1 - (K56_D * (u * cr * cr)) * s2;
    }
    if (!(x <= K56_XMAX)) return (x != x) ? x : 0.0;
    int e;
    (void)frexp(x, &e);                       // x in [2^(e-1), 2^e), e >= 1
    int idx = e - 1;
    double z;
    if (idx < 5) {
        z = ldexp(x, 1 - idx) - 3.0;          // 2 x / 2^idx - 3  in [-1, 1)
    } else {
        idx = 5;
        z = 64.0 / x - 1.0;                   // (-1, 1]
    }
    const double *c = cheb + idx * K56_NDEG;
    const double z2 = z + z;
    double b1 = 0.0, b2 = 0.0;
#pragma unroll
    for (int k = K56_NDEG - 1; k >= 1; --k) {
        const double b0 = fma(z2, b1, c[k]) - b2;
        b2 = b1;
        b1 = b0;
    }
    const double g = fma(z, b1, c[0]) - b2;
    return (K56_PRE * cbrt(u)) * exp(-x) * g;
}

__host__ __device__ __forceinline__ double vonkarman_unit(double u) { return vonkarman_unit_tab(u, &k56_cheb[0][0]); }

#ifdef __HIPCC__
// copy the table into `lds` (6 * K56_NDEG doubles); the caller synchronises the workgroup afterwards
__device__ __forceinline__ void vonkarman_stage_table(double *lds) {
    for (int i = threadIdx.x; i < 6 * K56_NDEG; i += blockDim.x) lds[i] = (&k56_cheb[0][0])[i];
}
#endif

