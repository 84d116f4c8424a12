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

__device__ __forceinline__ double quad_form(const KParams &p, double dx, double dy) {
    return p.a * dx * dx + p.b2 * dx * dy + p.c * dy * dy;
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
