// CPU-side AddressSanitizer / UBSan run of the pure host logic of libtgp.so (no GPU needed): Morton keys, counting sort,
// packed-layout helpers, tile maps, and the error paths of the C-ABI when there is no device.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "tgp.h"
void tgp_morton_keys(const double *x, const double *y, int64_t n, std::vector<uint32_t> &key, int &nbuckets);
void tgp_counting_sort_row(const int64_t *src, int64_t n, const std::vector<uint32_t> &key, int nbuckets, std::vector<int64_t> &count, int64_t *dst);
extern "C" int tgp_debug_tilemap(int64_t T, int32_t *ti, int32_t *tj, int64_t cap);
int main() {
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0, 1);
    for (int64_t n : {2, 3, 255, 256, 257, 5000, 70000}) {
        std::vector<double> x(n), y(n);
        for (auto &v : x) v = u(g);
        for (auto &v : y) v = u(g);
        if (n == 255) for (auto &v : x) v = 0.5;              // degenerate extent in x
        std::vector<uint32_t> key; int nb = 0;
        tgp_morton_keys(x.data(), y.data(), n, key, nb);
        std::vector<int64_t> count, dst(n), src(n);
        tgp_counting_sort_row(nullptr, n, key, nb, count, dst.data());
        std::vector<char> seen(n, 0);
        for (int64_t i = 0; i < n; ++i) { if (dst[i] < 0 || dst[i] >= n || seen[dst[i]]) { printf("bad permutation\n"); return 1; } seen[dst[i]] = 1; }
        for (int64_t i = 1; i < n; ++i) if (key[dst[i]] < key[dst[i - 1]]) { printf("not sorted\n"); return 1; }
        std::uniform_int_distribution<int64_t> r(0, n - 2 > 0 ? n - 2 : 0);
        for (auto &s : src) s = r(g);                           // a bootstrap row (with repeats)
        tgp_counting_sort_row(src.data(), n, key, nb, count, dst.data());
        for (int64_t i = 1; i < n; ++i) if (key[dst[i]] < key[dst[i - 1]]) { printf("resample not sorted\n"); return 1; }
    }
    for (int64_t Np : {256, 512, 4096, 65536, 131072}) {
        if (tgp_padded_n(Np - 1) != Np || tgp_panel_off(0, Np) != 0) { printf("layout\n"); return 1; }
        int64_t prev = -1;
        for (int64_t p = 0; p <= Np / 256; ++p) { int64_t o = tgp_panel_off(p, Np); if (o <= prev) { printf("panel_off not increasing\n"); return 1; } prev = o; }
        if (tgp_panel_off(Np / 256, Np) != tgp_panel_elems(Np)) { printf("panel_elems\n"); return 1; }
    }
    for (int64_t T : {1, 7, 31, 32, 100, 191, 192, 496}) {
        std::vector<int32_t> ti(400000), tj(400000);
        int g2 = tgp_debug_tilemap(T, ti.data(), tj.data(), 400000);
        if (g2 <= 0) { printf("tilemap grid\n"); return 1; }
        std::vector<char> hit(T * T, 0); int64_t cnt = 0;
        for (int b = 0; b < g2; ++b) if (ti[b] >= 0) { if (tj[b] > ti[b] || ti[b] >= T || hit[ti[b] * T + tj[b]]) { printf("tilemap dup/out of range\n"); return 1; } hit[ti[b] * T + tj[b]] = 1; ++cnt; }
        if (cnt != T * (T + 1) / 2) { printf("tilemap incomplete %lld\n", (long long)cnt); return 1; }
    }
    tgp_ctx *ctx = nullptr; int dev = 0;
    int rc = tgp_init(&dev, 1, &ctx);
    printf("host logic ok under sanitizers; tgp_init without a device -> %d (device count %d)\n", rc, tgp_device_count());
    if (rc == 0) tgp_destroy(ctx);
    return 0;
}
