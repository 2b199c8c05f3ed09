// C++ host-side mirror (include/ws_watershed.hpp) exercised the way the reference's README
// quickstart and unit tests use the Rust API.  `test_mirror cpu` needs no GPU; `test_mirror gpu`
// runs the quickstart on device 0 and checks it against the CPU oracle (test infrastructure).
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/ws_watershed.hpp"
#include "../../oracle/ws_oracle.h"

namespace ws = rustronomy_watershed;

#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

static int cpu_checks() {
  // lib.rs:936-946 defaults, lib.rs:1026-1030 validation
  try {
    ws::TransformBuilder<>().set_max_water_lvl(255).build_segmenting();
    CHECK(false);
  } catch (const ws::BuildErr &e) {
    CHECK(e.kind == ws::BuildErr::MaxToHigh && e.value == 255);
  } catch (...) { CHECK(false); }
  try {
    ws::TransformBuilder<>().set_max_water_lvl(0).build_merging();
    CHECK(false);
  } catch (const ws::BuildErr &e) {
    CHECK(e.kind == ws::BuildErr::MaxToLow && e.value == 0);
  } catch (...) { CHECK(false); }
  CHECK(ws::UNCOLOURED == 0 && ws::NORMAL_MAX == 254 && ws::ALWAYS_FILL == 0 && ws::NEVER_FILL == 255);
  ws_options o;
  CHECK(ws_options_default(&o) == WS_OK && o.max_water_level == 254 && o.edge_correction == 0);
  std::printf("cpu checks ok\n");
  return 0;
}

static int no_device_check() {
  try {
    auto w = ws::TransformBuilder<>().build_segmenting();
    (void)w;
    std::printf("a device is present\n");
  } catch (const ws::WatershedError &e) {
    CHECK(e.status == WS_ERR_NO_DEVICE);
    std::printf("no device: fails loudly (%s)\n", e.what());
  }
  return 0;
}

static int gpu_checks() {
  const size_t H = 300, W = 420;
  ws::Array2<uint8_t> rf(H, W);
  ws_or_random_field(rf.data.data(), H, W, 7);                       // README.md:60: Uniform(0,254)
  auto watershed = ws::TransformBuilder<>().build_segmenting();      // README.md:62
  auto mins = watershed.find_local_minima(rf);                       // README.md:64
  std::vector<uint64_t> want_rc(2 * H * W);
  const size_t n = ws_or_find_local_minima(rf.data.data(), H, W, want_rc.data(), H * W);
  CHECK(n == mins.size());
  for (size_t i = 0; i < n; ++i) CHECK(mins[i].first == want_rc[2 * i] && mins[i].second == want_rc[2 * i + 1]);
  auto out = watershed.transform(rf, mins);                          // README.md:66
  std::vector<uint64_t> want(H * W);
  CHECK(ws_or_segment(rf.data.data(), H, W, want_rc.data(), n, 254, 0, WS_OR_TIE_FIRST, 0, want.data(), nullptr, nullptr,
                      nullptr, nullptr, nullptr) == 0);
  CHECK(out.rows == H && out.cols == W);
  CHECK(std::memcmp(out.data.data(), want.data(), H * W * 8) == 0);

  // hook: count coloured pixels per level (HookCtx, lib.rs:844-862) vs the oracle's hook
  struct Acc { std::vector<size_t> v; } acc;
  auto cb = [](void *u, uint8_t, uint8_t, const uint8_t *, const uint64_t *lab, size_t h, size_t w) {
    size_t c = 0;
    for (size_t i = 0; i < h * w; ++i) c += lab[i] != 0;
    static_cast<Acc *>(u)->v.push_back(c);
  };
  CHECK(ws_or_segment(rf.data.data(), H, W, want_rc.data(), n, 40, 1, WS_OR_TIE_FIRST, 0, std::vector<uint64_t>((H + 2) * (W + 2)).data(),
                      nullptr, nullptr, cb, &acc, nullptr) == 0);
  auto hooked = ws::TransformBuilder<size_t>()
                    .set_max_water_lvl(40)
                    .enable_edge_correction()
                    .set_wlvl_hook([](const ws::HookCtx &c) {
                      size_t k = 0;
                      for (size_t r = 0; r < c.colours.rows; ++r)
                        for (size_t q = 0; q < c.colours.cols; ++q) k += c.colours(r, q) != 0;
                      return k;
                    })
                    .build_segmenting();
  auto counts = hooked.transform_with_hook(rf, mins);
  CHECK(counts.size() == 41 && counts == acc.v);

  // merging transform_to_list (tests/core_bench.rs:48 shape, small): conservation per level
  auto merging = ws::TransformBuilder<>().set_max_water_lvl(60).build_merging();
  auto list = merging.transform_to_list(rf, mins);
  CHECK(list.size() == 61);
  for (auto &lv : list) {
    CHECK(lv.second.size() == H * W + 1);
    uint64_t s = 0;
    for (uint64_t a : lv.second) s += a;
    CHECK(s == H * W);
  }
  // out-of-bounds seed: the reference panics (lib.rs:1676)
  try {
    watershed.transform(rf, {{H, 0}});
    CHECK(false);
  } catch (const std::out_of_range &) {}
  std::printf("gpu checks ok: %zu seeds, labels bit-exact vs oracle\n", n);
  return 0;
}

int main(int argc, char **argv) {
  const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
  if (cpu_checks()) return 1;
  if (gpu) return gpu_checks();
  return no_device_check();
}
