// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access shapes the library's kernels use
// (MI355X_MICROARCH.md, HBM section: 16-B/lane streaming reads report exactly half; other widths must be calibrated on a
// known byte count in one's own access pattern).  Every kernel moves exactly BYTES bytes once, from a buffer far larger
// than the 256 MiB Infinity Cache:
//   read16   16 B per lane, consecutive lanes consecutive        (operand loads of the trailing update)
//   read8    8 B per lane, consecutive lanes consecutive
//   read8t   8 B per lane in the MFMA accumulator shape: 16 lanes cover 128 B of one row, the wave's four lane groups
//            four different rows 2 KiB apart                      (the C preload of the trailing update)
//   write8 / write16  the same widths as streaming stores         (C write-back / K build)
// run:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./fetch_calib ; likewise WRITE_SIZE
// build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/fetch_calib tools/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>

static const size_t BYTES = 4ull << 30;

__global__ void read16(const double2 *p, double *sink, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += st) { double2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) *sink = acc;
}
__global__ void read8(const double *p, double *sink, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    double acc = 0;
    for (; i < n; i += st) acc += p[i];
    if (acc == 1.2345e300) *sink = acc;
}
// rows of 256 doubles (2 KiB); a wave reads a 4-row x 16-column patch per instruction: lane = (row = lane / 16, col = lane % 16)
__global__ void read8t(const double *p, double *sink, size_t nrows) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t patch = (blockIdx.x * (size_t)(blockDim.x >> 6) + wave), npatch = nrows / 4 * 16, st = (size_t)gridDim.x * (blockDim.x >> 6);
    double acc = 0;
    for (; patch < npatch; patch += st) {
        size_t r4 = patch / 16, c16 = patch % 16;
        acc += p[(r4 * 4 + lane / 16) * 256 + c16 * 16 + lane % 16];
    }
    if (acc == 1.2345e300) *sink = acc;
}
__global__ void write16(double2 *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = make_double2(1.0, 2.0);
}
__global__ void write8(double *p, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += st) p[i] = 3.0;
}

int main() {
    double *d, *sink;
    if (hipMalloc(&d, BYTES) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 0, BYTES);
    const int grid = 256 * 16, block = 256;
    for (int rep = 0; rep < 2; ++rep) {
        read16<<<grid, block>>>((const double2 *)d, sink, BYTES / 16);
        read8<<<grid, block>>>(d, sink, BYTES / 8);
        read8t<<<grid, block>>>(d, sink, BYTES / 2048);
        write16<<<grid, block>>>((double2 *)d, BYTES / 16);
        write8<<<grid, block>>>(d, BYTES / 8);
    }
    hipDeviceSynchronize();
    printf("each kernel moved %zu bytes\n", BYTES);
    return 0;
}
