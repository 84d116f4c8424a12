// Store-only ceilings for the K build's access shapes (no arithmetic): what HBM write rate does this box sustain for
//   stream16   grid-stride 16 B/lane stores (4 KiB per workgroup and iteration)
//   slab       workgroup b writes bytes [128 KiB b, 128 KiB (b+1)): 64 rows of 2 KiB, wave w rows w, w+4, ..., two 1 KiB wave stores per row
//   slab_nt    the same with nontemporal stores
//   tile       128 x 128 tiles of a ld-256 panel (1 KiB wave stores, rows 2 KiB apart), the first-generation K build's shape
// at 4 GiB and 17 GB per launch.  build: hipcc --offload-arch=gfx950 -O2 -o /tmp/write_ceiling tools/probes/write_ceiling.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void stream16(double2 *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_double2(1.0, 2.0);
}
template <bool NT>
__global__ __launch_bounds__(256) void slab(double *p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *dst = p + (size_t)blockIdx.x * 64 * 256 + 2 * lane;
#pragma unroll 4
    for (int r = wave; r < 64; r += 4) {
        double *q = dst + (size_t)r * 256;
        if (NT) {
            __builtin_nontemporal_store(1.0, q); __builtin_nontemporal_store(2.0, q + 1);
            __builtin_nontemporal_store(1.0, q + 128); __builtin_nontemporal_store(2.0, q + 129);
        } else {
            *reinterpret_cast<double2 *>(q) = make_double2(1.0, 2.0);
            *reinterpret_cast<double2 *>(q + 128) = make_double2(1.0, 2.0);
        }
    }
}
__global__ __launch_bounds__(256) void tile(double *p) {     // workgroup b: rows 128 (b / 2) .., column half b & 1 of one long panel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *dst = p + (size_t)(blockIdx.x >> 1) * 128 * 256 + (blockIdx.x & 1) * 128 + 2 * lane;
#pragma unroll 4
    for (int r = wave; r < 128; r += 4) *reinterpret_cast<double2 *>(dst + (size_t)r * 256) = make_double2(1.0, 2.0);
}

// the slab shape with NF fp64 FMAs of make-work per stored element (4 independent chains per lane and row): how much arithmetic hides
// under the store stream?
template <int NF>
__global__ __launch_bounds__(256) void slab_valu(double *p, double seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *dst = p + (size_t)blockIdx.x * 64 * 256 + 2 * lane;
    const double c = seed * 1e-3 + lane * 1e-6;
#pragma unroll 4
    for (int r = wave; r < 64; r += 4) {
        double v[4] = {seed + r, seed - r, seed * r, seed + 2.0 * r};
#pragma unroll
        for (int k = 0; k < NF; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fma(v[e], c, 0.5 + k);
        double *q = dst + (size_t)r * 256;
        *reinterpret_cast<double2 *>(q) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2 *>(q + 128) = make_double2(v[2], v[3]);
    }
}

// the K build's own arithmetic on the slab shape, piece by piece: Gaussian kernel value from coordinates (ROWLOAD: the row's
// coordinates come from wave-uniform loads as in the library, else they are made up from the row number), DIAG: the diagonal /
// padding selects of the library kernel evaluated for every element
__device__ __forceinline__ double pexp2_neg(double s) {
    const double magic = 6755399441055744.0;
    const double t = -fmin(s, 1021.0);
    const double z = t + magic;
    const double k = z - magic;
    const double f = t - k;
    double p = 1.3691488853904128881e-12;
    p = fma(p, f, 2.5678435993488205142e-11); p = fma(p, f, 4.4455382718708114976e-10); p = fma(p, f, 7.0549116208011233299e-9);
    p = fma(p, f, 1.0178086009239699727e-7); p = fma(p, f, 1.3215486790144309488e-6); p = fma(p, f, 1.525273380405984028e-5);
    p = fma(p, f, 1.5403530393381609954e-4); p = fma(p, f, 1.3333558146428443423e-3); p = fma(p, f, 9.618129107628477162e-3);
    p = fma(p, f, 5.5504108664821579953e-2); p = fma(p, f, 2.4022650695910071233e-1); p = fma(p, f, 6.9314718055994530942e-1);
    p = fma(p, f, 1.0);
    return __hiloint2double(__double2hiint(p) + (__double2loint(z) << 20), __double2loint(p));
}
template <bool ROWLOAD, bool DIAG>
__global__ __launch_bounds__(256) void slab_k(double *p, const double *__restrict__ X, long n, double a, double b2, double c, double amp) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    double *dst = p + (size_t)blockIdx.x * 64 * 256 + 2 * lane;
    const long i0 = ((long)blockIdx.x * 64) % (n - 64), j0 = ((long)blockIdx.x * 37) % (n - 256);
    long jc[2] = {j0 + 2 * lane, j0 + 128 + 2 * lane};
    double xj[2][2], yj[2][2];
    for (int h = 0; h < 2; ++h)
        for (int e = 0; e < 2; ++e) { xj[h][e] = X[2 * (jc[h] + e)]; yj[h][e] = X[2 * (jc[h] + e) + 1]; }
#pragma unroll 4
    for (int r = wave; r < 64; r += 4) {
        const long i = i0 + r;
        const double xi = ROWLOAD ? X[2 * i] : 1e-5 * (double)i, yi = ROWLOAD ? X[2 * i + 1] : 3e-5 * (double)i;
        double2 v[2];
        for (int h = 0; h < 2; ++h) {
            double dx = xi - xj[h][0], dy = yi - yj[h][0];
            v[h].x = amp * pexp2_neg(a * dx * dx + b2 * dx * dy + c * dy * dy);
            dx = xi - xj[h][1]; dy = yi - yj[h][1];
            v[h].y = amp * pexp2_neg(a * dx * dx + b2 * dx * dy + c * dy * dy);
            if (DIAG) {
                if (i == jc[h]) v[h].x = amp + 1e-3;
                if (i == jc[h] + 1) v[h].y = amp + 1e-3;
                if (jc[h] >= n) v[h].x = 0.0;
                if (jc[h] + 1 >= n) v[h].y = 0.0;
            }
        }
        double *q = dst + (size_t)r * 256;
        *reinterpret_cast<double2 *>(q) = v[0];
        *reinterpret_cast<double2 *>(q + 128) = v[1];
    }
}

template <typename F>
static void timeit(const char *name, size_t bytes, F launch) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("%-10s %6.2f GB  %7.3f ms  %7.1f GB/s  %.1f %% of 8 TB/s\n", name, bytes / 1e9, best, bytes / best / 1e6, bytes / best / 1e6 / 80.0);
}

int main() {
    const size_t big = (size_t)17180000000ull / (128 * 1024) * (128 * 1024);
    double *d;
    if (hipMalloc(&d, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
    const long npts = 65536;
    double *dX;
    hipMalloc(&dX, npts * 16);
    {
        double *h = (double *)malloc(npts * 16);
        for (long i = 0; i < 2 * npts; ++i) h[i] = (double)rand() / RAND_MAX;
        hipMemcpy(dX, h, npts * 16, hipMemcpyHostToDevice);
        free(h);
    }
    for (size_t bytes : {(size_t)4 << 30, big}) {
        timeit("stream16", bytes, [&] { stream16<<<256 * 16, 256>>>((double2 *)d, bytes / 16); });
        timeit("slab", bytes, [&] { slab<false><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d); });
        timeit("slab_nt", bytes, [&] { slab<true><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d); });
        timeit("tile", bytes, [&] { tile<<<(unsigned)(bytes / (128 * 1024)), 256>>>(d); });
        timeit("k:noload", bytes, [&] { slab_k<false, false><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, dX, npts, 144.0, 57.0, 220.0, 1.0); });
        timeit("k:rowload", bytes, [&] { slab_k<true, false><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, dX, npts, 144.0, 57.0, 220.0, 1.0); });
        timeit("k:row+diag", bytes, [&] { slab_k<true, true><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, dX, npts, 144.0, 57.0, 220.0, 1.0); });
        timeit("slab+8fma", bytes, [&] { slab_valu<8><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, 1.25); });
        timeit("slab+16fma", bytes, [&] { slab_valu<16><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, 1.25); });
        timeit("slab+24fma", bytes, [&] { slab_valu<24><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, 1.25); });
        timeit("slab+32fma", bytes, [&] { slab_valu<32><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, 1.25); });
        timeit("slab+48fma", bytes, [&] { slab_valu<48><<<(unsigned)(bytes / (128 * 1024)), 256>>>(d, 1.25); });
    }
    return 0;
}
