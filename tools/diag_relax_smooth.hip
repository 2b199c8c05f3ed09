// diag_relax_smooth.hip -- diagnostic build of the relaxation on a SMOOTH field (never shipped): per-pass time and, for the
// workgroups of chosen passes, where a tile run spends its time (s_memrealtime stamps, 10 ns).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DWS_DIAG_STAMPS -Irustronomy-watershed_amd/csrc -o tools/_build/diag_smooth tools/diag_relax_smooth.hip
// usage: diag_smooth [corr=64] [N=8192]
#include "../rustronomy-watershed_amd/csrc/ws_relax.hip"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace wsk;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ unsigned long long mix64d(unsigned long long x) { unsigned long long z = x + 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
__device__ float lattice(int i, int j) { return (float)(mix64d(((unsigned long long)(unsigned)i << 32) | (unsigned)j) >> 40) / 16777216.0f; }
// value noise, Catmull-Rom in both directions: smooth like the bicubic low-pass noise of tools/exp_smooth.py
__device__ float cr(float a, float b, float c, float d, float t) { return b + 0.5f * t * (c - a + t * (2 * a - 5 * b + 4 * c - d + t * (3 * (b - c) + d - a))); }
__global__ void k_img(uint8_t *img, int W, int H, int corr) {
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= W) return;
  const float fx = (x + 0.5f) / corr, fy = (y + 0.5f) / corr; const int ix = (int)fx, iy = (int)fy; const float tx = fx - ix, ty = fy - iy;
  float col[4];
  for (int j = 0; j < 4; ++j) col[j] = cr(lattice(ix - 1, iy - 1 + j), lattice(ix, iy - 1 + j), lattice(ix + 1, iy - 1 + j), lattice(ix + 2, iy - 1 + j), tx);
  float v = cr(col[0], col[1], col[2], col[3], ty);
  v = fminf(fmaxf((v + 0.2f) / 1.4f, 0.f), 1.f);
  img[(size_t)y * W + x] = (uint8_t)(v * 253.0f);
}
__global__ void k_seedkeys(const uint8_t *img, uint32_t *keys, int H, int W) {
  int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y; if (x >= W) return; uint32_t k = KEY_INF;
  if (y >= 1 && y < H - 1 && x >= 1 && x < W - 1) { uint8_t v = img[(size_t)y * W + x]; bool ok = true;
    for (int dy = -1; dy <= 1; ++dy) for (int dx = -1; dx <= 1; ++dx) if ((dy || dx) && img[(size_t)(y + dy) * W + x + dx] >= v) ok = false;
    if (ok) k = 0; }
  keys[(size_t)y * W + x] = k;
}
int main(int argc, char **argv) {
  const int corr = argc > 1 ? atoi(argv[1]) : 64, N = argc > 2 ? atoi(argv[2]) : 8192;
  const int H = N, W = N; const size_t n = (size_t)H * W;
  uint8_t *img; uint32_t *keys, *stamps, *flags, *tile_list; unsigned long long *diag;
  const int ntiles = (int)relax_tiles(H, W);
  const size_t flag_words = (size_t)(COUNTER_RING + 4) * FLAG_SLOT;
  CHECK(hipMalloc(&img, n)); CHECK(hipMalloc(&keys, n * 4)); CHECK(hipMalloc(&stamps, (size_t)ntiles * 8 * 4)); CHECK(hipMalloc(&flags, flag_words * 4));
  CHECK(hipMalloc(&diag, (size_t)ntiles * 8 * 8)); CHECK(hipMalloc(&tile_list, relax_list_words(H, W) * 4));
  k_img<<<dim3((W + 255) / 256, H), 256>>>(img, W, H, corr);
  k_seedkeys<<<dim3((W + 255) / 256, H), 256>>>(img, keys, H, W);
  CHECK(hipMemset(stamps, 0, (size_t)ntiles * 8 * 4)); CHECK(hipMemset(flags, 0, flag_words * 4)); CHECK(hipMemset(tile_list, 0, relax_list_words(H, W) * 4));
  CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), &diag, sizeof(diag)));
  PassFlags pf{flags, flags + COUNTER_RING * FLAG_SLOT, flags + (COUNTER_RING + 3) * FLAG_SLOT, flags + (COUNTER_RING + 1) * FLAG_SLOT};
  std::vector<uint32_t> slot(FLAG_SLOT), st(2 * FLAG_SLOT);
  std::vector<unsigned long long> h((size_t)ntiles * 8);
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  double total = 0;
  for (uint32_t pass = 0; pass < 2000; ++pass) {
    CHECK(hipMemset(diag, 0, (size_t)ntiles * 64)); CHECK(hipMemset(flags + (COUNTER_RING + 1) * FLAG_SLOT, 0, 2 * FLAG_SLOT * 4));
    CHECK(hipEventRecord(a));
    CHECK(relax_pass(0, img, W, keys, H, W, 254, pass, stamps, pf, 0xFFFFFFFFu, nullptr, false, 0, false, false, tile_list));
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); total += ms;
    CHECK(hipMemcpy(slot.data(), flags + (pass % COUNTER_RING) * FLAG_SLOT, FLAG_SLOT * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(st.data(), flags + (COUNTER_RING + 1) * FLAG_SLOT, 2 * FLAG_SLOT * 4, hipMemcpyDeviceToHost));
    unsigned long long tiles = 0, rounds = 0; for (int i = 0; i < FLAG_SLOT; ++i) { tiles += st[i]; rounds += st[FLAG_SLOT + i]; }
    bool any = false; for (auto v : slot) any |= v != 0;
    if (pass >= 7 && (pass % 8 == 0 || !any)) {
      CHECK(hipMemcpy(h.data(), diag, h.size() * 8, hipMemcpyDeviceToHost));
      std::vector<double> load, loop, tail, rows, cols, chk; unsigned long long t_first = ~0ull, t_last = 0; int nwg = 0;
      for (int t = 0; t < ntiles; ++t) { auto *p = &h[(size_t)t * 8]; if (!p[3]) continue; ++nwg; load.push_back((p[1] - p[0]) / 100.0); loop.push_back((p[2] - p[1]) / 100.0);
        tail.push_back((p[3] - p[2]) / 100.0); rows.push_back(p[5] / 100.0); cols.push_back(p[6] / 100.0); chk.push_back(p[7] / 100.0); t_first = std::min(t_first, p[0]); t_last = std::max(t_last, p[3]); }
      auto pct = [](std::vector<double> v, double q) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
      printf("pass %4u: %7.1f us  tiles %6llu rounds %6llu | last tile run of %4d workgroups, us p50/p90: load %.1f/%.1f  rounds %.1f/%.1f (row scans %.1f/%.1f  col scans %.1f/%.1f  checked sweeps %.1f/%.1f)  epilogue %.1f/%.1f\n",
             pass, ms * 1e3, tiles, rounds, nwg, pct(load, .5), pct(load, .9), pct(loop, .5), pct(loop, .9), pct(rows, .5), pct(rows, .9), pct(cols, .5), pct(cols, .9), pct(chk, .5), pct(chk, .9), pct(tail, .5), pct(tail, .9));
    }
    if (!any) { printf("converged after pass %u, %.2f ms of passes (with a host round trip per pass)\n", pass, total); break; }
  }
  return 0;
}
