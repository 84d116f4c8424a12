// Covariance-function evaluation on device (fp64).
//   treegp/kernels.py:114-126  AnisotropicRBF  exp(-0.5 d^2), d^2 = dX^T invLam dX
//   treegp/kernels.py:249-276  VonKarman       (d/l)^(5/6) K_{5/6}(2 pi d/l) / lim0, 1 at d == 0
//   treegp/kernels.py:355-381  AnisotropicVonKarman, same with the Mahalanobis distance
// sklearn's Product(ConstantKernel(amp), k) is folded in as one multiply after the
// transcendental (amp * k), the order NumPy evaluates it in.
#pragma once
#include <hip/hip_runtime.h>
#include "bessel_k56.h"
#include "../../include/tgp.h"

struct KParams {
    double amp, a, b2, c, inv_ell;
    int kind;
};

static inline KParams make_kparams(const tgp_kernel *k) {
    KParams p;
    p.amp = k->amp;
    p.a = k->a;
    p.b2 = 2.0 * k->b;
    p.c = k->c;
    p.inv_ell = (k->ell != 0.0) ? 1.0 / k->ell : 0.0;
    p.kind = k->kind;
    return p;
}

enum { KE_GAUSS = 0, KE_VK = 1, KE_AVK = 2 };

static inline int kind_to_ke(int kind) {
    switch (kind) {
        case TGP_RBF:
        case TGP_ARBF: return KE_GAUSS;
        case TGP_VK: return KE_VK;
        case TGP_AVK: return KE_AVK;
        default: return -1;
    }
}

// 2^(-s) for s >= 0, <= 1 ulp
__device__ __forceinline__ double tgp_exp2_neg(double s) {     // 2^(-s), s >= 0; Taylor of 2^f, |f| <= 0.5, degree 13
    // fp64 adds / FMAs and one 32-bit integer op: rint, the int conversion and ldexp of the textbook form are replaced by
    // the 1.5 * 2^52 rounding constant (the low dword of t + magic IS round(t) in two's complement) and an add into the
    // exponent field.  Used by the K build (kbuild.hip: the probe tools/probes/write_ceiling.hip puts the arithmetic that
    // hides under its store stream at ~32 FMA-equivalents per element; this is 18 + 8 for the quadratic form and the
    // amplitude).  It was tried in the hope that rint / cvt / ldexp were slow instructions; they are not (the fused predict
    // keeps the textbook form, which is one instruction shorter), so the K build's rate did not move with it either.
    // s is clamped at 1021 so that the result stays a normal number: beyond it the true value is below 4.5e-308 and
    // 2^-1021 p stands in for it (absolute error < 4.5e-308).  A negative s (an indefinite invLam) is clamped at -1023: the
    // exponent-field add below must not run into the infinity / NaN encodings; K then holds numbers of order 1e308 and the
    // factorisation reports it as not positive definite.  A NaN s (NaN coordinate or parameter) stays NaN -- fmin / fmax
    // would drop it and the point would silently count as uncorrelated; SciPy's cholesky (check_finite) raises there.
    const double magic = 6755399441055744.0;
    const double t = (s != s) ? s : -fmax(fmin(s, 1021.0), -1023.0);
    const double z = t + magic;
    const double k = z - magic;
    const double f = t - k;
    double p = 1.3691488853904128881e-12;
    p = fma(p, f, 2.5678435993488205142e-11);
    p = fma(p, f, 4.4455382718708114976e-10);
    p = fma(p, f, 7.0549116208011233299e-9);
    p = fma(p, f, 1.0178086009239699727e-7);
    p = fma(p, f, 1.3215486790144309488e-6);
    p = fma(p, f, 1.525273380405984028e-5);
    p = fma(p, f, 1.5403530393381609954e-4);
    p = fma(p, f, 1.3333558146428443423e-3);
    p = fma(p, f, 9.618129107628477162e-3);
    p = fma(p, f, 5.5504108664821579953e-2);
    p = fma(p, f, 2.4022650695910071233e-1);
    p = fma(p, f, 6.9314718055994530942e-1);
    p = fma(p, f, 1.0);
    const int ki = __double2loint(z);                       // round(t), -1021 .. 1023 (t = NaN: p is NaN, whatever ki)
    return __hiloint2double(__double2hiint(p) + (ki << 20), __double2loint(p));
}

__device__ __forceinline__ double quad_form(const KParams &p, double dx, double dy) {
    return p.a * dx * dx + p.b2 * dx * dy + p.c * dy * dy;
}

// the same with the von Karman Chebyshev table read from `tab` (an LDS copy, bessel_k56.h); tab is ignored for Gaussians
template <int KE>
__device__ __forceinline__ double kernel_value_tab(const KParams &p, double dx, double dy, const double *tab) {
    if constexpr (KE == KE_GAUSS) {
        return p.amp * exp(-0.5 * (p.a * dx * dx + p.b2 * dx * dy + p.c * dy * dy));
    } else if constexpr (KE == KE_VK) {
        return p.amp * vonkarman_unit_tab(sqrt(dx * dx + dy * dy) * p.inv_ell, tab);
    } else {
        return p.amp * vonkarman_unit_tab(sqrt(p.a * dx * dx + p.b2 * dx * dy + p.c * dy * dy), tab);
    }
}

// value WITHOUT the amplitude; coincident points give exactly 1
template <int KE>
__device__ __forceinline__ double kernel_unit(const KParams &p, double dx, double dy) {
    if constexpr (KE == KE_GAUSS) {
        return exp(-0.5 * quad_form(p, dx, dy));
    } else if constexpr (KE == KE_VK) {
        double u = sqrt(dx * dx + dy * dy) * p.inv_ell;
        return vonkarman_unit(u);
    } else {
        double u = sqrt(quad_form(p, dx, dy));
        return vonkarman_unit(u);
    }
}

template <int KE>
__device__ __forceinline__ double kernel_value(const KParams &p, double dx, double dy) {
    return p.amp * kernel_unit<KE>(p, dx, dy);
}
