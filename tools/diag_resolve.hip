// diag_resolve.hip -- diagnostic build of k_resolve_local with per-workgroup phase stamps (never shipped).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_DIAG_STAMPS -Irustronomy-watershed_amd/csrc -o tools/_build/diag_resolve tools/diag_resolve.hip
#include "../rustronomy-watershed_amd/csrc/ws_relax.hip"
#include "../rustronomy-watershed_amd/csrc/ws_kernels.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace wsk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ unsigned long long mix64d(unsigned long long x) { unsigned long long z = x + 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void k_img(uint8_t *img, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) img[i] = (uint8_t)(mix64d((1ull << 40) + i) % 254u); }
// seeds = strict 8-neighbour maxima, painted with an arbitrary colour (index + 1)
__global__ void k_seedplanes(const uint8_t *img, uint32_t *keys, uint32_t *labels, int H, int W) {
  int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= W) return; uint32_t k = KEY_INF, l = 0;
  if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) { uint8_t v = img[(size_t)y * W + x]; bool ok = true;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if ((dy || dx) && img[(size_t)(y + dy) * W + x + dx] >= v) ok = false;
    if (ok) { k = 0; l = (uint32_t)((size_t)y * W + x) + 1u; } }
  keys[(size_t)y * W + x] = k; labels[(size_t)y * W + x] = l;
}
int main() {
  const int H = 8192, W = 8192; const size_t n = (size_t)H * W;
  uint8_t *img; uint32_t *keys, *labels, *stamps, *flags, *refs; unsigned long long *diag;
  const int ntiles = (int)relax_tiles(H, W), rtiles = (int)resolve_tiles(H, W);
  const size_t flag_words = (size_t)(COUNTER_RING + 6) * FLAG_SLOT;
  CHECK(hipMalloc(&img, n)); CHECK(hipMalloc(&keys, n * 4)); CHECK(hipMalloc(&labels, n * 4)); CHECK(hipMalloc(&stamps, (size_t)ntiles * 8 * 4));
  CHECK(hipMalloc(&flags, flag_words * 4)); CHECK(hipMalloc(&refs, resolve_ref_capacity(H, W) * 4));
  CHECK(hipMalloc(&diag, (size_t)rtiles * 8 * 8));
  k_img<<<(n + 255) / 256, 256>>>(img, n);
  for (int rep = 0; rep < 3; ++rep) {
    k_seedplanes<<<dim3(W / 256, H), 256>>>(img, keys, labels, H, W);
    CHECK(hipMemset(stamps, 0, (size_t)ntiles * 8 * 4)); CHECK(hipMemset(flags, 0, flag_words * 4));
    unsigned long long *none = nullptr; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &none, sizeof(none)));
    PassFlags pf{flags, flags + COUNTER_RING * FLAG_SLOT, flags + (COUNTER_RING + 3) * FLAG_SLOT, nullptr};
    for (uint32_t pass = 0; pass < 12; ++pass) CHECK(relax_pass(0, img, W, keys, H, W, 254, pass, stamps, pf, 0xFFFFFFFFu));   // 12 passes: converged on this field
    CHECK(hipMemset(diag, 0, (size_t)rtiles * 64));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &diag, sizeof(diag)));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); CHECK(hipEventRecord(a));
    CHECK(resolve_two_launch(0, keys, labels, H, W, refs));
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)rtiles * 8); CHECK(hipMemcpy(h.data(), diag, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ph[5], life; double rounds = 0; unsigned long long t_first = ~0ull, t_last = 0;
    for (int t = 0; t < rtiles; ++t) { auto *p = &h[(size_t)t * 8];
      for (int k = 0; k < 5; ++k) ph[k].push_back((p[k + 1] - p[k]) / 100.0);
      life.push_back((p[5] - p[0]) / 100.0); rounds += p[6]; t_first = std::min(t_first, p[0]); t_last = std::max(t_last, p[5]); }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    printf("resolve (local+chase) %.1f us (event), local span %.1f us | per-WG us p50/p90: load %.1f/%.1f  parents %.1f/%.1f  jumping %.1f/%.1f  colours+store %.1f/%.1f  reflist %.1f/%.1f | life %.1f/%.1f | rounds avg %.2f\n",
           ms * 1e3, (t_last - t_first) / 100.0, pct(ph[0], .5), pct(ph[0], .9), pct(ph[1], .5), pct(ph[1], .9), pct(ph[2], .5), pct(ph[2], .9),
           pct(ph[3], .5), pct(ph[3], .9), pct(ph[4], .5), pct(ph[4], .9), pct(life, .5), pct(life, .9), rounds / rtiles);
  }
  return 0;
}
