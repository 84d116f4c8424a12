// Which bits of HW_REG_HW_ID / HW_REG_XCC_ID identify a compute unit on gfx950?  Launches 512 workgroups of 256 threads
// that stay resident for a while (so that all of them are on the chip at once) and prints the distinct values of every field.
// build: hipcc --offload-arch=gfx950 -O2 -o gpurun_out/hwid_probe tools/probes/hwid_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>

__global__ void probe(unsigned *out, int spin) {
    extern __shared__ double dyn[];
    if (spin < 0) dyn[threadIdx.x] = 1.0;            // keeps the dynamic LDS allocation
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
        out[blockIdx.x * 4 + 2] = (unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffffffu);
    }
    long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < spin) {}
}

int main(int argc, char **argv) {
    const int nwg = 512;
    const int lds = argc > 1 ? atoi(argv[1]) : 0;         // dynamic LDS per workgroup: how many of them share a CU?
    if (lds > 0) hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    unsigned *d;
    hipMalloc(&d, nwg * 16);
    hipMemset(d, 0, nwg * 16);
    probe<<<nwg, 256, lds>>>(d, 2000000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(nwg * 4);
    hipMemcpy(h.data(), d, nwg * 16, hipMemcpyDeviceToHost);
    printf("first 16 workgroups: blockIdx  HW_ID       XCC_ID\n");
    for (int i = 0; i < 16; ++i) printf("  %3d  0x%08x  0x%08x\n", i, h[i * 4], h[i * 4 + 1]);
    struct F { const char *name; int reg, lo, n; };
    const F fields[] = {{"wave_id[3:0]", 0, 0, 4}, {"simd_id[5:4]", 0, 4, 2}, {"pipe_id[7:6]", 0, 6, 2}, {"cu_id[11:8]", 0, 8, 4},
                        {"sh_id[12]", 0, 12, 1}, {"se_id[15:13]", 0, 13, 3}, {"tg_id[19:16]", 0, 16, 4}, {"vm_id[23:20]", 0, 20, 4},
                        {"queue_id[26:24]", 0, 24, 3}, {"state[29:27]", 0, 27, 3}, {"me_id[31:30]", 0, 30, 2},
                        {"xcc_id[3:0]", 1, 0, 4}, {"xcc rest[31:4]", 1, 4, 28}};
    for (const F &f : fields) {
        std::set<unsigned> v;
        for (int i = 0; i < nwg; ++i) v.insert((h[i * 4 + f.reg] >> f.lo) & ((f.n == 32 ? 0u : (1u << f.n)) - 1u));
        printf("%-18s %zu distinct:", f.name, v.size());
        int c = 0;
        for (unsigned x : v) { if (c++ < 20) printf(" %u", x); }
        printf("\n");
    }
    // key used by the queue kernel: HW_ID[15:8] within an XCC
    std::map<unsigned, int> per;
    int agree = 0;
    for (int i = 0; i < nwg; ++i) {
        const unsigned key = ((h[i * 4 + 1] & 15u) << 8) | ((h[i * 4] >> 8) & 0xffu);
        per[key]++;
        agree += ((h[i * 4 + 1] & 7u) == (unsigned)(i & 7));
    }
    printf("distinct (xcc, HW_ID[15:8]) keys: %zu for %d resident workgroups; xcc_id == blockIdx & 7 for %d of them\n", per.size(), nwg, agree);
    {   // workgroups of the first round (started within 100 us of the first one) per compute unit
        unsigned t0 = 0xffffffffu;
        for (int i = 0; i < nwg; ++i) t0 = h[i * 4 + 2] < t0 ? h[i * 4 + 2] : t0;
        std::map<unsigned, int> first;
        int nfirst = 0;
        for (int i = 0; i < nwg; ++i)
            if (h[i * 4 + 2] - t0 < 10000u) { first[((h[i * 4 + 1] & 15u) << 8) | ((h[i * 4] >> 8) & 0xffu)]++; ++nfirst; }
        std::map<int, int> hh;
        for (auto &kv : first) hh[kv.second]++;
        printf("dynamic LDS %d B: %d workgroups resident in the first round on %zu CUs:", lds, nfirst, first.size());
        for (auto &kv : hh) printf("  %d CUs x %d", kv.second, kv.first);
        printf("\n");
    }
    std::map<int, int> hist;
    for (auto &kv : per) hist[kv.second]++;
    for (auto &kv : hist) printf("  keys with %d workgroups: %d\n", kv.first, kv.second);
    return 0;
}
