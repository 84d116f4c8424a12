// What does a hand-off between two RESIDENT workgroups cost on gfx950 -- the primitive a persistent, dependency-counted
// factorisation kernel would be made of?  Two workgroups ping-pong a 128 x 128 fp64 tile (128 KiB) through global memory
// `iters` times: write the tile, release (agent scope: L2 write-back), set a flag; the partner polls the flag, acquires (L2
// invalidate), reads the tile.  Pairs on the SAME XCD (blockIdx 0 and 8) and on DIFFERENT XCDs (blockIdx 0 and 1); a second
// series hands over only the flag (no payload).  Every spin is bounded: a lost hand-off ends the kernel with an error
// code instead of hanging the GPU.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/handoff_probe tools/probes/handoff_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define TILE (128 * 128)
#define SPIN_MAX (1 << 22)

__device__ __forceinline__ bool wait_flag(const unsigned *flag, unsigned want) {
    for (int s = 0; s < SPIN_MAX; ++s) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// me = 0 / 1; tiles[me] is written by me, tiles[1 - me] by the partner; flags[me] counts my completed writes
__global__ __launch_bounds__(256) void pingpong(double *tiles, unsigned *flags, unsigned long long *out, int iters, int partner_block,
                                                int payload) {
    int me;
    if (blockIdx.x == 0) me = 0;
    else if ((int)blockIdx.x == partner_block) me = 1;
    else return;
    __shared__ int ok;
    double *mine = tiles + (size_t)me * TILE, *theirs = tiles + (size_t)(1 - me) * TILE;
    const int tid = threadIdx.x;
    double acc = 0.0;
    if (tid == 0) ok = 1;
    __syncthreads();
    unsigned long long t0 = 0;
    if (me == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 1; it <= iters; ++it) {
        if (me == 1 || it > 1) {                     // wait for the partner's hand-off number it (me == 1) / it - 1 (me == 0)
            if (tid == 0) {
                if (!wait_flag(flags + (1 - me), me == 1 ? (unsigned)it : (unsigned)(it - 1))) ok = 0;
            }
            __syncthreads();
            if (!ok) break;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            if (payload)
                for (int i = tid; i < TILE; i += 256) acc += theirs[i];
        }
        if (payload)
            for (int i = tid; i < TILE; i += 256) mine[i] = acc * 1e-30 + (double)(it + i);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + me, (unsigned)it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (me == 0) {                                   // the last answer
        if (tid == 0 && ok && !wait_flag(flags + 1, (unsigned)iters)) ok = 0;
        __syncthreads();
        if (tid == 0) {
            out[0] = __builtin_amdgcn_s_memrealtime() - t0;
            out[1] = (unsigned long long)ok;
            out[2] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
        }
    } else if (tid == 0) {
        out[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
        out[4] = (unsigned long long)ok;
    }
    if (acc == 12345.678) out[5] = 1;                // keeps the loads
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    double *tiles;
    unsigned *flags;
    unsigned long long *out, h[8];
    hipMalloc(&tiles, 2 * TILE * sizeof(double));
    hipMalloc(&flags, 64);
    hipMalloc(&out, 64);
    for (int payload = 1; payload >= 0; --payload)
        for (int partner : {8, 1}) {
            hipMemset(flags, 0, 64);
            hipMemset(out, 0, 64);
            hipMemset(tiles, 0, 2 * TILE * sizeof(double));
            pingpong<<<16, 256>>>(tiles, flags, out, iters, partner, payload);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
            // s_memrealtime ticks at 100 MHz; one iteration = two hand-offs (there and back)
            printf("%s, partner block %d (XCC %llu -> %llu): %s  %.2f us per hand-off (%d iterations, ok %llu/%llu)\n",
                   payload ? "128 KiB tile + flag" : "flag only", partner, h[2], h[3], hipGetErrorString(e),
                   h[0] * 0.01 / (2.0 * iters), iters, h[1], h[4]);
        }
    return 0;
}
