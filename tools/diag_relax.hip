// diag_relax.hip -- diagnostic build of k_relax with per-workgroup phase stamps (never shipped).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_DIAG_STAMPS -Irustronomy-watershed_amd/csrc -o tools/_build/diag tools/diag_relax.hip
#include "../rustronomy-watershed_amd/csrc/ws_relax.hip"
#include <algorithm>
#include <cstdio>
#include <vector>
using namespace wsk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ unsigned long long mix64d(unsigned long long x) { unsigned long long z = x + 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__global__ void k_img(uint8_t *img, size_t n) { size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; if (i < n) img[i] = (uint8_t)(mix64d((1ull << 40) + i) % 254u); }
__global__ void k_seedkeys(const uint8_t *img, uint32_t *keys, int H, int W) {
  int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= W) return; uint32_t k = KEY_INF;
  if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) { uint8_t v = img[(size_t)y * W + x]; bool ok = true;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if ((dy || dx) && img[(size_t)(y + dy) * W + x + dx] >= v) ok = false;
    if (ok) k = 0; }
  keys[(size_t)y * W + x] = k;
}
int main() {
  const int H = 8192, W = 8192; const size_t n = (size_t)H * W;
  uint8_t *img; uint32_t *keys, *stamps, *flags; unsigned long long *diag;
  const int ntiles = (int)relax_tiles(H, W);
  const size_t flag_words = (size_t)(COUNTER_RING + 4) * FLAG_SLOT;
  CHECK(hipMalloc(&img, n)); CHECK(hipMalloc(&keys, n * 4)); CHECK(hipMalloc(&stamps, (size_t)ntiles * 8 * 4)); CHECK(hipMalloc(&flags, flag_words * 4));
  CHECK(hipMalloc(&diag, (size_t)ntiles * 8 * 8));
  k_img<<<(n + 255) / 256, 256>>>(img, n);
  for (int rep = 0; rep < 4; ++rep) {
    const bool with_stats = rep >= 2;
    k_seedkeys<<<dim3(W / 256, H), 256>>>(img, keys, H, W);
    CHECK(hipMemset(stamps, 0, (size_t)ntiles * 8 * 4)); CHECK(hipMemset(flags, 0, flag_words * 4)); CHECK(hipMemset(diag, 0, (size_t)ntiles * 64));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &diag, sizeof(diag)));
    PassFlags pf{flags, flags + COUNTER_RING * FLAG_SLOT, flags + (COUNTER_RING + 3) * FLAG_SLOT, with_stats ? flags + (COUNTER_RING + 1) * FLAG_SLOT : nullptr};
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b)); CHECK(hipEventRecord(a));
    CHECK(relax_pass(0, img, W, keys, H, W, 254, 0, stamps, pf, 0xFFFFFFFFu));
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    std::vector<unsigned long long> h((size_t)ntiles * 8); CHECK(hipMemcpy(h.data(), diag, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> load, loop, tail, life; std::vector<unsigned long long> it; unsigned long long t_first = ~0ull, t_last = 0;
    for (int t = 0; t < ntiles; ++t) { auto *p = &h[(size_t)t * 8]; load.push_back((p[1] - p[0]) / 100.0); loop.push_back((p[2] - p[1]) / 100.0); tail.push_back((p[3] - p[2]) / 100.0);
      life.push_back((p[3] - p[0]) / 100.0); it.push_back(p[4]); t_first = std::min(t_first, p[0]); t_last = std::max(t_last, p[3]); }
    auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
    double itsum = 0; for (auto x : it) itsum += x;
    printf("stats=%d pass0 %.1f us (event)  span %.1f us | per-WG us: load p50 %.1f p90 %.1f | loop p50 %.1f p90 %.1f | store p50 %.1f p90 %.1f | life p50 %.1f p90 %.1f | iters avg %.2f max %llu\n",
           (int)with_stats, ms * 1e3, (t_last - t_first) / 100.0, pct(load, .5), pct(load, .9), pct(loop, .5), pct(loop, .9), pct(tail, .5), pct(tail, .9), pct(life, .5), pct(life, .9),
           itsum / ntiles, *std::max_element(it.begin(), it.end()));
  }
  return 0;
}
