// Fused GP mean prediction (seam S3): ys[q] = sum_i amp k(Xs_q, X_i) alpha_i, i.e.
// kernel(X2, Y=X1) followed by HT @ alpha (treegp/gp_interp.py:177,183) without ever
// materialising the (M, N) cross-kernel matrix.  fp64-VALU-bound (one transcendental per pair);
// HBM traffic is just the coordinates.
//
// One thread per query point; the training points (x, y, alpha) are staged through LDS in
// tiles of 256 and read back as wave-wide broadcasts.  The training set is split over
// gridDim.y so that small M still fills the chip; partial sums are combined in a fixed order
// by a second kernel (bitwise reproducible, no atomics).
#include "tgp_internal.h"
#include "kernel_eval.h"

namespace {
constexpr int PT = 256;   // training points per LDS tile

template <int KE>
__global__ __launch_bounds__(256) void predict_partial_kernel(KParams p, const double *__restrict__ X, int64_t n,
                                                              const double *__restrict__ alpha,
                                                              const double *__restrict__ Xs, int64_t m,
                                                              double *__restrict__ partial, int64_t chunk) {
    __shared__ double sx[PT], sy[PT], sa[PT];
    __shared__ double vk_tab[KE == KE_GAUSS ? 1 : 6 * K56_NDEG];       // von Karman: the Chebyshev table, gathered per lane
    if constexpr (KE != KE_GAUSS) vonkarman_stage_table(vk_tab);       // (the first barrier of the loop below publishes it)
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk;
    const int64_t i_end = (i_begin + chunk < n) ? i_begin + chunk : n;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = Xs[2 * q]; yq = Xs[2 * q + 1]; }
    double acc = 0.0;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += PT) {
        const int64_t i = i0 + tid;
        __syncthreads();
        if (i < i_end) { sx[tid] = X[2 * i]; sy[tid] = X[2 * i + 1]; sa[tid] = alpha[i]; }
        else { sx[tid] = 0.0; sy[tid] = 0.0; sa[tid] = 0.0; }
        __syncthreads();
        const int cnt = (int)((i_end - i0 < PT) ? (i_end - i0) : PT);
        if (cnt == PT) {
#pragma unroll 4
            for (int t = 0; t < PT; ++t) acc += kernel_value_tab<KE>(p, xq - sx[t], yq - sy[t], vk_tab) * sa[t];
        } else {
            for (int t = 0; t < cnt; ++t) acc += kernel_value_tab<KE>(p, xq - sx[t], yq - sy[t], vk_tab) * sa[t];
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc;
}

// ---- Gaussian fast path ---------------------------------------------------------------------------
// exp(-0.5 d^T invLam d) with invLam = L L^T is 2^-(|u|^2) for u = sqrt(0.5 log2 e) L^T d.  Coordinates
// are transformed once per call (O(n + m)), which leaves 2 sub + 1 mul + 1 fma for the exponent, and
// 2^-s needs no range checks for s >= 0: 22 fp64 instructions per pair instead of 32.  The rounding of
// the exponent differs from the reference's a dx^2 + 2 b dx dy + c dy^2 by ~1e-16 |q| relative
// (parity tests: 1e-10 on predicted values).  The amplitude is applied in the reduction.
__global__ __launch_bounds__(256) void predict_transform_kernel(const double *__restrict__ X, int64_t n, double t00,
                                                                double t10, double t11, double *__restrict__ U) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double x = X[2 * i], y = X[2 * i + 1];
    U[2 * i] = t00 * x + t10 * y;
    U[2 * i + 1] = t11 * y;
}

// (the K build's variant, tgp_exp2_neg in kernel_eval.h -- magic-constant rounding and an integer add into the exponent instead of
// rint / cvt / ldexp, plus a clamp -- is one instruction longer and measured 5 % slower here: 12.6 vs 12.0 ms per 1.7e10 pairs;
// v_rndne_f64, v_cvt_i32_f64 and v_ldexp_f64 issue at the full fp64 rate on gfx950)
__device__ __forceinline__ double exp2_neg(double s) {        // 2^(-s), s >= 0
    const double t = -s;
    const double k = rint(t);
    const double f = t - k;                                    // exact, |f| <= 0.5
    double p = 1.3691488853904128881e-12;                      // Taylor of 2^f = sum (f ln2)^i / i!, i <= 13
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
    return ldexp(p, (int)k);
}

// Round 3: 2^t as 2^e 2^(j/NT) 2^(r/NT) with t NT = e NT + j + r, |r| <= 1/2: 2^(j/NT) from an NT-entry table in LDS (NT = 32:
// 256 bytes, every entry in banks of its own, so a wave's gather is conflict-free whatever the lanes ask for), 2^(r/NT) from a
// degree-6 (NT = 32) or degree-5 (NT = 64) Taylor polynomial (remainder 3.5e-18 / 3.5e-17 relative).  The coordinates are
// pre-scaled by sqrt(NT) as well, so the squared distance IS t NT.  19 (18) VALU instructions per pair instead of 22:
// 2 sub, mul, fma | rndne, sub | 6 (5) fma | cvt, and, shift (table address), shift (exponent) | mul, ldexp | fma.
// Table entries are correctly rounded (mpmath), the product T p adds one rounding: <= 1.5 ulp.
// 2^(j/256), j = 0 .. 255, correctly rounded (mpmath); the 32- and 64-entry tables are every 8th / 4th entry
__constant__ double EXP2_J256[256] = {
    0x1.0000000000000p+0, 0x1.00b1afa5abcbfp+0, 0x1.0163da9fb3335p+0, 0x1.02168143b0281p+0, 0x1.02c9a3e778061p+0, 0x1.037d42e11bbccp+0,
    0x1.04315e86e7f85p+0, 0x1.04e5f72f654b1p+0, 0x1.059b0d3158574p+0, 0x1.0650a0e3c1f89p+0, 0x1.0706b29ddf6dep+0, 0x1.07bd42b72a836p+0,
    0x1.0874518759bc8p+0, 0x1.092bdf66607e0p+0, 0x1.09e3ecac6f383p+0, 0x1.0a9c79b1f3919p+0, 0x1.0b5586cf9890fp+0, 0x1.0c0f145e46c85p+0,
    0x1.0cc922b7247f7p+0, 0x1.0d83b23395decp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.0efa55fdfa9c5p+0, 0x1.0fb66affed31bp+0, 0x1.1073028d7233ep+0,
    0x1.11301d0125b51p+0, 0x1.11edbab5e2ab6p+0, 0x1.12abdc06c31ccp+0, 0x1.136a814f204abp+0, 0x1.1429aaea92de0p+0, 0x1.14e95934f312ep+0,
    0x1.15a98c8a58e51p+0, 0x1.166a45471c3c2p+0, 0x1.172b83c7d517bp+0, 0x1.17ed48695bbc0p+0, 0x1.18af9388c8deap+0, 0x1.1972658375d2fp+0,
    0x1.1a35beb6fcb75p+0, 0x1.1af99f8138a1cp+0, 0x1.1bbe084045cd4p+0, 0x1.1c82f95281c6bp+0, 0x1.1d4873168b9aap+0, 0x1.1e0e75eb44027p+0,
    0x1.1ed5022fcd91dp+0, 0x1.1f9c18438ce4dp+0, 0x1.2063b88628cd6p+0, 0x1.212be3578a819p+0, 0x1.21f49917ddc96p+0, 0x1.22bdda27912d1p+0,
    0x1.2387a6e756238p+0, 0x1.2451ffb82140ap+0, 0x1.251ce4fb2a63fp+0, 0x1.25e85711ece75p+0, 0x1.26b4565e27cddp+0, 0x1.2780e341ddf29p+0,
    0x1.284dfe1f56381p+0, 0x1.291ba7591bb70p+0, 0x1.29e9df51fdee1p+0, 0x1.2ab8a66d10f13p+0, 0x1.2b87fd0dad990p+0, 0x1.2c57e39771b2fp+0,
    0x1.2d285a6e4030bp+0, 0x1.2df961f641589p+0, 0x1.2ecafa93e2f56p+0, 0x1.2f9d24abd886bp+0, 0x1.306fe0a31b715p+0, 0x1.31432edeeb2fdp+0,
    0x1.32170fc4cd831p+0, 0x1.32eb83ba8ea32p+0, 0x1.33c08b26416ffp+0, 0x1.3496266e3fa2dp+0, 0x1.356c55f929ff1p+0, 0x1.36431a2de883bp+0,
    0x1.371a7373aa9cbp+0, 0x1.37f26231e754ap+0, 0x1.38cae6d05d866p+0, 0x1.39a401b7140efp+0, 0x1.3a7db34e59ff7p+0, 0x1.3b57fbfec6cf4p+0,
    0x1.3c32dc313a8e5p+0, 0x1.3d0e544ede173p+0, 0x1.3dea64c123422p+0, 0x1.3ec70df1c5175p+0, 0x1.3fa4504ac801cp+0, 0x1.40822c367a024p+0,
    0x1.4160a21f72e2ap+0, 0x1.423fb2709468ap+0, 0x1.431f5d950a897p+0, 0x1.43ffa3f84b9d4p+0, 0x1.44e086061892dp+0, 0x1.45c2042a7d232p+0,
    0x1.46a41ed1d0057p+0, 0x1.4786d668b3237p+0, 0x1.486a2b5c13cd0p+0, 0x1.494e1e192aed2p+0, 0x1.4a32af0d7d3dep+0, 0x1.4b17dea6db7d7p+0,
    0x1.4bfdad5362a27p+0, 0x1.4ce41b817c114p+0, 0x1.4dcb299fddd0dp+0, 0x1.4eb2d81d8abffp+0, 0x1.4f9b2769d2ca7p+0, 0x1.508417f4531eep+0,
    0x1.516daa2cf6642p+0, 0x1.5257de83f4eefp+0, 0x1.5342b569d4f82p+0, 0x1.542e2f4f6ad27p+0, 0x1.551a4ca5d920fp+0, 0x1.56070dde910d2p+0,
    0x1.56f4736b527dap+0, 0x1.57e27dbe2c4cfp+0, 0x1.58d12d497c7fdp+0, 0x1.59c0827ff07ccp+0, 0x1.5ab07dd485429p+0, 0x1.5ba11fba87a03p+0,
    0x1.5c9268a5946b7p+0, 0x1.5d84590998b93p+0, 0x1.5e76f15ad2148p+0, 0x1.5f6a320dceb71p+0, 0x1.605e1b976dc09p+0, 0x1.6152ae6cdf6f4p+0,
    0x1.6247eb03a5585p+0, 0x1.633dd1d1929fdp+0, 0x1.6434634ccc320p+0, 0x1.652b9febc8fb7p+0, 0x1.6623882552225p+0, 0x1.671c1c70833f6p+0,
    0x1.68155d44ca973p+0, 0x1.690f4b19e9538p+0, 0x1.6a09e667f3bcdp+0, 0x1.6b052fa75173ep+0, 0x1.6c012750bdabfp+0, 0x1.6cfdcddd47645p+0,
    0x1.6dfb23c651a2fp+0, 0x1.6ef9298593ae5p+0, 0x1.6ff7df9519484p+0, 0x1.70f7466f42e87p+0, 0x1.71f75e8ec5f74p+0, 0x1.72f8286ead08ap+0,
    0x1.73f9a48a58174p+0, 0x1.74fbd35d7cbfdp+0, 0x1.75feb564267c9p+0, 0x1.77024b1ab6e09p+0, 0x1.780694fde5d3fp+0, 0x1.790b938ac1cf6p+0,
    0x1.7a11473eb0187p+0, 0x1.7b17b0976cfdbp+0, 0x1.7c1ed0130c132p+0, 0x1.7d26a62ff86f0p+0, 0x1.7e2f336cf4e62p+0, 0x1.7f3878491c491p+0,
    0x1.80427543e1a12p+0, 0x1.814d2add106d9p+0, 0x1.82589994cce13p+0, 0x1.8364c1eb941f7p+0, 0x1.8471a4623c7adp+0, 0x1.857f4179f5b21p+0,
    0x1.868d99b4492edp+0, 0x1.879cad931a436p+0, 0x1.88ac7d98a6699p+0, 0x1.89bd0a478580fp+0, 0x1.8ace5422aa0dbp+0, 0x1.8be05bad61778p+0,
    0x1.8cf3216b5448cp+0, 0x1.8e06a5e0866d9p+0, 0x1.8f1ae99157736p+0, 0x1.902fed0282c8ap+0, 0x1.9145b0b91ffc6p+0, 0x1.925c353aa2fe2p+0,
    0x1.93737b0cdc5e5p+0, 0x1.948b82b5f98e5p+0, 0x1.95a44cbc8520fp+0, 0x1.96bdd9a7670b3p+0, 0x1.97d829fde4e50p+0, 0x1.98f33e47a22a2p+0,
    0x1.9a0f170ca07bap+0, 0x1.9b2bb4d53fe0dp+0, 0x1.9c49182a3f090p+0, 0x1.9d674194bb8d5p+0, 0x1.9e86319e32323p+0, 0x1.9fa5e8d07f29ep+0,
    0x1.a0c667b5de565p+0, 0x1.a1e7aed8eb8bbp+0, 0x1.a309bec4a2d33p+0, 0x1.a42c980460ad8p+0, 0x1.a5503b23e255dp+0, 0x1.a674a8af46052p+0,
    0x1.a799e1330b358p+0, 0x1.a8bfe53c12e59p+0, 0x1.a9e6b5579fdbfp+0, 0x1.ab0e521356ebap+0, 0x1.ac36bbfd3f37ap+0, 0x1.ad5ff3a3c2774p+0,
    0x1.ae89f995ad3adp+0, 0x1.afb4ce622f2ffp+0, 0x1.b0e07298db666p+0, 0x1.b20ce6c9a8952p+0, 0x1.b33a2b84f15fbp+0, 0x1.b468415b749b1p+0,
    0x1.b59728de5593ap+0, 0x1.b6c6e29f1c52ap+0, 0x1.b7f76f2fb5e47p+0, 0x1.b928cf22749e4p+0, 0x1.ba5b030a1064ap+0, 0x1.bb8e0b79a6f1fp+0,
    0x1.bcc1e904bc1d2p+0, 0x1.bdf69c3f3a207p+0, 0x1.bf2c25bd71e09p+0, 0x1.c06286141b33dp+0, 0x1.c199bdd85529cp+0, 0x1.c2d1cd9fa652cp+0,
    0x1.c40ab5fffd07ap+0, 0x1.c544778fafb22p+0, 0x1.c67f12e57d14bp+0, 0x1.c7ba88988c933p+0, 0x1.c8f6d9406e7b5p+0, 0x1.ca3405751c4dbp+0,
    0x1.cb720dcef9069p+0, 0x1.ccb0f2e6d1675p+0, 0x1.cdf0b555dc3fap+0, 0x1.cf3155b5bab74p+0, 0x1.d072d4a07897cp+0, 0x1.d1b532b08c968p+0,
    0x1.d2f87080d89f2p+0, 0x1.d43c8eacaa1d6p+0, 0x1.d5818dcfba487p+0, 0x1.d6c76e862e6d3p+0, 0x1.d80e316c98398p+0, 0x1.d955d71ff6075p+0,
    0x1.da9e603db3285p+0, 0x1.dbe7cd63a8315p+0, 0x1.dd321f301b460p+0, 0x1.de7d5641c0658p+0, 0x1.dfc97337b9b5fp+0, 0x1.e11676b197d17p+0,
    0x1.e264614f5a129p+0, 0x1.e3b333b16ee12p+0, 0x1.e502ee78b3ff6p+0, 0x1.e653924676d76p+0, 0x1.e7a51fbc74c83p+0, 0x1.e8f7977cdb740p+0,
    0x1.ea4afa2a490dap+0, 0x1.eb9f4867cca6ep+0, 0x1.ecf482d8e67f1p+0, 0x1.ee4aaa2188510p+0, 0x1.efa1bee615a27p+0, 0x1.f0f9c1cb6412ap+0,
    0x1.f252b376bba97p+0, 0x1.f3ac948dd7274p+0, 0x1.f50765b6e4540p+0, 0x1.f6632798844f8p+0, 0x1.f7bfdad9cbe14p+0, 0x1.f91d802243c89p+0,
    0x1.fa7c1819e90d8p+0, 0x1.fbdba3692d514p+0, 0x1.fd3c22b8f71f1p+0, 0x1.fe9d96b2a23d9p+0};

template <int NT>
__device__ __forceinline__ double exp2_neg_tab(double w, const double *tab) {      // 2^(-w / NT), w >= 0
    constexpr double L = 0.69314718055994530942 / NT;
    const double t = -w;
    const double k = rint(t);
    const double r = t - k;                                    // exact, |r| <= 0.5
    double p;
    if constexpr (NT == 32) {
        p = L * L * L * L * L * L / 720.0;
        p = fma(p, r, L * L * L * L * L / 120.0);
        p = fma(p, r, L * L * L * L / 24.0);
    } else if constexpr (NT == 64) {
        p = L * L * L * L * L / 120.0;
        p = fma(p, r, L * L * L * L / 24.0);
    } else {
        p = L * L * L * L / 24.0;
    }
    p = fma(p, r, L * L * L / 6.0);
    p = fma(p, r, L * L / 2.0);
    p = fma(p, r, L);
    p = fma(p, r, 1.0);
    const int ki = (int)k;                                     // saturating conversion: huge distances end as ldexp(., -2^26) = 0
    if constexpr (NT == 256) {
        // table address = 8 * (low byte of ki) in ONE instruction (SDWA byte select) instead of and + shift
        unsigned off;
        asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(off) : "v"(3u), "v"(ki));
        const double T = *reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab) + off);
        return ldexp(T * p, ki >> 8);
    }
    return ldexp(tab[ki & (NT - 1)] * p, ki >> (NT == 32 ? 5 : (NT == 64 ? 6 : 8)));
}

template <int NT>
__global__ __launch_bounds__(256) void predict_gauss_tab_kernel(const double *__restrict__ U, int64_t n,
                                                                const double *__restrict__ alpha,
                                                                const double *__restrict__ Us, int64_t m,
                                                                double *__restrict__ partial, int64_t chunk) {
    __shared__ double sx[PT], sy[PT], sa[PT];
    __shared__ double tab[NT];
    const int tid = threadIdx.x;
    if (tid < NT) tab[tid] = EXP2_J256[tid * (256 / NT)];
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk;
    const int64_t i_end = (i_begin + chunk < n) ? i_begin + chunk : n;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = Us[2 * q]; yq = Us[2 * q + 1]; }
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += PT) {
        const int64_t i = i0 + tid;
        __syncthreads();
        if (i < i_end) { sx[tid] = U[2 * i]; sy[tid] = U[2 * i + 1]; sa[tid] = alpha[i]; }
        else { sx[tid] = 0.0; sy[tid] = 0.0; sa[tid] = 0.0; }          // alpha = 0: padded entries add nothing
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < PT; t += 2) {
            const double dx0 = xq - sx[t], dy0 = yq - sy[t];
            const double dx1 = xq - sx[t + 1], dy1 = yq - sy[t + 1];
            acc0 = fma(exp2_neg_tab<NT>(fma(dx0, dx0, dy0 * dy0), tab), sa[t], acc0);
            acc1 = fma(exp2_neg_tab<NT>(fma(dx1, dx1, dy1 * dy1), tab), sa[t + 1], acc1);
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc0 + acc1;
}

// (Round 2's kernel, TGP_PREDICT_EXP=0.)  Where its rate comes from: the inner loop is 22 fp64 VALU instructions per pair (13 fma of the polynomial, 3 add, 2 fmac, rndne,
// cvt, ldexp, mul) and nothing else but one ds_read_b128 per 1.3 pairs, i.e. 256 CUs x 4 SIMDs x 16 lanes x f / 22 pairs/s:
// 1.79e12 at 2.4 GHz, 1.64e12 at the 2.2 GHz (1.25 kW) the chip settles at under this kernel; measured back to back 1.50e12
// (tools/predict_clock.py), 1.38 - 1.43e12 as one launch of a bench step.  Two
// queries per thread (half the LDS reads per pair) changed nothing (3.10 vs 3.14 ms at N = 32 768, M = 131 072): issue-bound.
__global__ __launch_bounds__(256) void predict_gauss_fast_kernel(const double *__restrict__ U, int64_t n,
                                                                 const double *__restrict__ alpha,
                                                                 const double *__restrict__ Us, int64_t m,
                                                                 double *__restrict__ partial, int64_t chunk) {
    __shared__ double sx[PT], sy[PT], sa[PT];
    const int tid = threadIdx.x;
    const int64_t q = (int64_t)blockIdx.x * 256 + tid;
    const int64_t i_begin = (int64_t)blockIdx.y * chunk;
    const int64_t i_end = (i_begin + chunk < n) ? i_begin + chunk : n;
    double xq = 0.0, yq = 0.0;
    if (q < m) { xq = Us[2 * q]; yq = Us[2 * q + 1]; }
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t i0 = i_begin; i0 < i_end; i0 += PT) {
        const int64_t i = i0 + tid;
        __syncthreads();
        if (i < i_end) { sx[tid] = U[2 * i]; sy[tid] = U[2 * i + 1]; sa[tid] = alpha[i]; }
        else { sx[tid] = 0.0; sy[tid] = 0.0; sa[tid] = 0.0; }          // alpha = 0: padded entries add nothing
        __syncthreads();
#pragma unroll 4
        for (int t = 0; t < PT; t += 2) {
            const double dx0 = xq - sx[t], dy0 = yq - sy[t];
            const double dx1 = xq - sx[t + 1], dy1 = yq - sy[t + 1];
            acc0 = fma(exp2_neg(fma(dx0, dx0, dy0 * dy0)), sa[t], acc0);
            acc1 = fma(exp2_neg(fma(dx1, dx1, dy1 * dy1)), sa[t + 1], acc1);
        }
    }
    if (q < m) partial[(int64_t)blockIdx.y * m + q] = acc0 + acc1;
}

__global__ __launch_bounds__(256) void predict_reduce_kernel(const double *__restrict__ partial, int64_t m, int nsplit,
                                                             double scale, double *__restrict__ ys) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= m) return;
    double s = 0.0;
    for (int k = 0; k < nsplit; ++k) s += partial[(int64_t)k * m + q];
    ys[q] = scale * s;
}
}  // namespace

int launch_predict(tgp_ctx *ctx, const tgp_kernel *k, const double *d_X, int64_t n, const double *d_alpha,
                   const double *d_Xs, int64_t m, double *d_ys) {
    const int ke = kind_to_ke(k->kind);
    TGP_ARG(ke >= 0);
    TGP_ARG(n > 0 && m > 0);
    const KParams p = make_kparams(k);
    const int64_t qblocks = (m + 255) / 256;
    // Many more workgroups than slots (256 CUs x 7 resident at 70 VGPRs = 1792), but never split finer than one LDS tile: with
    // 2048 equal workgroups the last 256 ran one per CU after the other 1792 had finished (9.06 ms at N = 65 536 / M = 262 144);
    // 3584 / 7168 / 14 336 / 28 672 workgroups: 8.54 / 8.39 / 8.28 / 8.27 ms = 2.07e12 pairs/s (profiles/r03_predict_exp.txt).
    // Forcing 8 waves per SIMD instead (41 VGPRs, everything resident at once) is slower: 9.9 ms.
    static const int64_t wg_target = [] { const char *e = getenv("TGP_PREDICT_WGS"); return e ? (int64_t)atoll(e) : (int64_t)14336; }();
    int64_t nsplit = (wg_target + qblocks - 1) / qblocks;
    const int64_t maxsplit = (n + PT - 1) / PT;
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    int64_t chunk = (n + nsplit - 1) / nsplit;
    chunk = (chunk + PT - 1) / PT * PT;
    nsplit = (n + chunk - 1) / chunk;
    auto rup = [](size_t b) { return (b + 255) / 256 * 256; };
    int rc = tgp_ensure_scratch(ctx, rup((size_t)nsplit * m * 8) + rup(2 * n * 8) + rup(2 * m * 8));
    if (rc) return rc;
    double *partial = (double *)ctx->scratch;
    dim3 grid((unsigned)qblocks, (unsigned)nsplit), block(256);
    static const bool no_fast = getenv("TGP_PREDICT_GENERIC") != nullptr;
    // invLam = L L^T (2x2 Cholesky); 1-D kernels have c = b = 0, i.e. l11 = 0
    const double l00 = (k->a > 0.0) ? sqrt(k->a) : 0.0;
    const double l10 = (l00 > 0.0) ? k->b / l00 : 0.0;
    const double d11 = k->c - l10 * l10;
    if (ke == KE_GAUSS && !no_fast && l00 > 0.0 && d11 >= 0.0) {
        // TGP_PREDICT_EXP: 256 (default) / 64 / 32 = table-driven 2^t with that many entries, 0 = the degree-13 polynomial of round 2
        static const int exp_tab = [] { const char *e = getenv("TGP_PREDICT_EXP"); return e ? atoi(e) : 256; }();
        const double sc = 0.84932180028801904272 * (exp_tab == 32 ? 5.6568542494923801952 : (exp_tab == 64 ? 8.0 : (exp_tab == 256 ? 16.0 : 1.0)));   // sqrt(0.5 log2 e NT)
        const double t00 = sc * l00, t10 = sc * l10, t11 = sc * sqrt(d11);
        double *U = (double *)((char *)ctx->scratch + rup((size_t)nsplit * m * 8));
        double *Us = (double *)((char *)U + rup(2 * n * 8));
        predict_transform_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(d_X, n, t00, t10, t11, U);
        predict_transform_kernel<<<(unsigned)((m + 255) / 256), 256, 0, ctx->stream>>>(d_Xs, m, t00, t10, t11, Us);
        if (exp_tab == 32) predict_gauss_tab_kernel<32><<<grid, block, 0, ctx->stream>>>(U, n, d_alpha, Us, m, partial, chunk);
        else if (exp_tab == 64) predict_gauss_tab_kernel<64><<<grid, block, 0, ctx->stream>>>(U, n, d_alpha, Us, m, partial, chunk);
        else if (exp_tab == 256) predict_gauss_tab_kernel<256><<<grid, block, 0, ctx->stream>>>(U, n, d_alpha, Us, m, partial, chunk);
        else predict_gauss_fast_kernel<<<grid, block, 0, ctx->stream>>>(U, n, d_alpha, Us, m, partial, chunk);
        predict_reduce_kernel<<<(unsigned)qblocks, 256, 0, ctx->stream>>>(partial, m, (int)nsplit, k->amp, d_ys);
        TGP_HIP(hipGetLastError());
        return 0;
    }
    switch (ke) {
        case KE_GAUSS: predict_partial_kernel<KE_GAUSS><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        case KE_VK: predict_partial_kernel<KE_VK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
        default: predict_partial_kernel<KE_AVK><<<grid, block, 0, ctx->stream>>>(p, d_X, n, d_alpha, d_Xs, m, partial, chunk); break;
    }
    predict_reduce_kernel<<<(unsigned)qblocks, 256, 0, ctx->stream>>>(partial, m, (int)nsplit, 1.0, d_ys);
    TGP_HIP(hipGetLastError());
    return 0;
}
