// The 16x16 Gauss-Jordan of potrf128.h (Cholesky factor L and D = L^-1 of one diagonal block in registers) timed on one wave:
//   v0  lanes 0..15, one DPP row, 15 DPP FMAs + the pivot row's scaling per column (gj16_dpp.h)
//   v1  the work over all four DPP rows: Schur columns replicated, the inverse's columns dealt to the rows, scaling deferred
//       (gj16s_dpp.h, gen_gj16s.py)
//   v2  the third form (gj16t_dpp.h, gen_gj16t.py): own-lane reciprocal square roots + one broadcast, nothing masked in the Schur
//       update, the inverse born from the identity
// Prints s_memtime ticks per 16x16 block and the largest deviation of L and D from a host fp64 reference.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I treegp_amd/csrc -o tools/probes/gj16_probe tools/probes/gj16_probe.hip
#include "potrf128.h"
#include <cstdio>
#include <cmath>
#include <vector>

using namespace potrf_v2;

template <int VARIANT>
__global__ __launch_bounds__(64) void probe(const double *A, double *Lout, double *Dout, unsigned long long *cyc, int reps) {
    __shared__ double M[16 * 17];
    __shared__ double Dl[16 * 17];
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) M[(e >> 4) * 17 + (e & 15)] = A[e];
    __syncthreads();
    double lsum = 0.0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < reps; ++it) {
        if constexpr (VARIANT == 0) {
            if (lane < 16) {
                const int i = lane;
                double t[16], ls[16];
#pragma clang loop unroll(full)
                for (int c = 0; c < 16; ++c) t[c] = (c <= i) ? M[i * 17 + c] : 0.0;
                int fail = -1;
                gauss_jordan16(t, ls, i, fail);
#pragma clang loop unroll(full)
                for (int c = 0; c < 16; ++c) {
                    Dl[i * 17 + c] = (c <= i) ? t[c] : 0.0;
                    if (it == reps - 1) { Dout[i * 16 + c] = (c <= i) ? t[c] : 0.0; Lout[i * 16 + c] = (c <= i) ? ls[c] : 0.0; }
                }
                lsum += ls[0];
            }
        } else {
            const int i = lane & 15, r = lane >> 4;
            double s[16], w[4] = {0.0, 0.0, 0.0, 0.0}, ls[16];
#pragma clang loop unroll(full)
            for (int c = 0; c < 16; ++c) s[c] = (c <= i || VARIANT == 2) ? M[i * 17 + c] : 0.0;
            int fail = -1;
            if constexpr (VARIANT == 2) gauss_jordan16t(s, w, ls, i, r, fail);
            else gauss_jordan16s(s, w, ls, i, r, fail);
#pragma clang loop unroll(full)
            for (int k = 0; k < 4; ++k) {
                const int c = 4 * k + r;
                Dl[i * 17 + c] = (c <= i) ? w[k] : 0.0;
                if (it == reps - 1) Dout[i * 16 + c] = (c <= i) ? w[k] : 0.0;
            }
            if (it == reps - 1 && r == 0) {
#pragma clang loop unroll(full)
                for (int c = 0; c < 16; ++c) Lout[i * 16 + c] = (c <= i) ? ls[c] : 0.0;
            }
            lsum += ls[0];
        }
        __builtin_amdgcn_s_waitcnt(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = (unsigned long long)(lsum != 12345.0); }
    if (lane == 1) Dout[256] = Dl[0];
}

int main() {
    std::vector<double> A(256), L(256, 0.0), D(256, 0.0);
    double B[16][16];                                        // SPD test block: B B^T + 4 I with a fixed pseudo-random B
    unsigned s = 12345u;
    for (auto &row : B) for (double &v : row) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double a = (i == j) ? 4.0 : 0.0; for (int k = 0; k < 16; ++k) a += B[i][k] * B[j][k]; A[i * 16 + j] = a; }
    for (int j = 0; j < 16; ++j) {                           // host reference: Cholesky, then forward substitution for the inverse
        double d = A[j * 16 + j];
        for (int k = 0; k < j; ++k) d -= L[j * 16 + k] * L[j * 16 + k];
        L[j * 16 + j] = std::sqrt(d);
        for (int i = j + 1; i < 16; ++i) { double v = A[i * 16 + j]; for (int k = 0; k < j; ++k) v -= L[i * 16 + k] * L[j * 16 + k]; L[i * 16 + j] = v / L[j * 16 + j]; }
    }
    for (int c = 0; c < 16; ++c) for (int i = c; i < 16; ++i) { double v = (i == c) ? 1.0 : 0.0; for (int k = c; k < i; ++k) v -= L[i * 16 + k] * D[k * 16 + c]; D[i * 16 + c] = v / L[i * 16 + i]; }
    double *dA, *dL, *dD; unsigned long long *dc;
    hipMalloc(&dA, 256 * 8); hipMalloc(&dL, 256 * 8); hipMalloc(&dD, 257 * 8); hipMalloc(&dc, 16);
    hipMemcpy(dA, A.data(), 256 * 8, hipMemcpyHostToDevice);
    const int reps = 2000;
    for (int variant = 0; variant < 3; ++variant) {
        for (int pass = 0; pass < 2; ++pass) {
            hipMemset(dL, 0, 256 * 8); hipMemset(dD, 0, 257 * 8);
            if (variant == 0) probe<0><<<1, 64>>>(dA, dL, dD, dc, reps); else if (variant == 1) probe<1><<<1, 64>>>(dA, dL, dD, dc, reps); else probe<2><<<1, 64>>>(dA, dL, dD, dc, reps);
            hipDeviceSynchronize();
        }
        std::vector<double> gL(256), gD(256); unsigned long long c[2];
        hipMemcpy(gL.data(), dL, 256 * 8, hipMemcpyDeviceToHost); hipMemcpy(gD.data(), dD, 256 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        double eL = 0, eD = 0;
        for (int e = 0; e < 256; ++e) { eL = std::fmax(eL, std::fabs(gL[e] - L[e])); eD = std::fmax(eD, std::fabs(gD[e] - D[e])); }
        printf("variant %d: %.0f s_memtime ticks per 16x16 block (load + Gauss-Jordan + store, %d repetitions); max |L - ref| %.2e, max |D - ref| %.2e\n",
               variant, (double)c[0] / reps, reps, eL, eD);
    }
    return 0;
}
