// C++ host-side mirror (include/ws_watershed.hpp) exercised the way the reference's README
// quickstart and unit tests use the Rust API.  `test_mirror cpu` needs no GPU; `test_mirror gpu`
// runs the quickstart on device 0 and checks it against the CPU oracle (test infrastructure).
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
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
  // the call pair as one call (ws_segment_minima): same labels, same list
  std::vector<ws::Seed> mins2;
  auto out2 = watershed.transform_from_minima(rf, &mins2);
  CHECK(mins2 == mins);
  CHECK(std::memcmp(out2.data.data(), want.data(), H * W * 8) == 0);
  CHECK(std::memcmp(watershed.transform_from_minima(rf).data.data(), want.data(), H * W * 8) == 0);
  // a cube of slices as one call (ws_segment_batch): slice k is the pair of calls on slice k
  {
    const size_t N = 5;
    std::vector<uint8_t> cube(N * H * W);
    for (size_t k = 0; k < N; ++k) ws_or_random_field(cube.data() + k * H * W, H, W, 20 + k);
    auto planes = watershed.transform_cube(cube.data(), N, H, W);
    CHECK(planes.size() == N);
    for (size_t k = 0; k < N; ++k) {
      std::vector<uint64_t> rc_k(2 * H * W), want_k(H * W);
      const size_t n_k = ws_or_find_local_minima(cube.data() + k * H * W, H, W, rc_k.data(), H * W);
      CHECK(ws_or_segment(cube.data() + k * H * W, H, W, rc_k.data(), n_k, 254, 0, WS_OR_TIE_FIRST, 0, want_k.data(), nullptr, nullptr,
                          nullptr, nullptr, nullptr) == 0);
      CHECK(planes[k].rows == H && planes[k].cols == W);
      CHECK(std::memcmp(planes[k].data.data(), want_k.data(), H * W * 8) == 0);
    }
  }

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

// The Rust shim (rust/src/watershed_hip.rs + shim.rs) cannot be compiled in this image.  This function makes the
// same C-ABI calls in the same order with the same arguments the shim makes for each public method, in plain C
// style, and checks every result against the oracle -- so the sequence is at least executed.  Comments name the
// Rust function each block stands for.
static int shim_sequence_checks() {
  const size_t H = 150, W = 212;
  std::vector<uint8_t> img(H * W);
  ws_or_random_field(img.data(), H, W, 11);
  // shim::HipCtx::new: ABI version check, then ws_ctx_create
  CHECK(ws_abi_version() == WS_ABI_VERSION);
  ws_ctx *ctx = nullptr;
  CHECK(ws_ctx_create(0, &ctx) == WS_OK);

  // WatershedUtils::find_local_minima: cap = one maximum per 2x2 block
  const size_t cap = ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
  std::vector<uint64_t> rc_pairs(2 * cap);
  size_t n = 0;
  CHECK(ws_find_local_minima(ctx, img.data(), H, W, W, rc_pairs.data(), cap, &n) == WS_OK);
  std::vector<uint64_t> want_rc(2 * H * W);
  CHECK(ws_or_find_local_minima(img.data(), H, W, want_rc.data(), H * W) == n);
  CHECK(std::memcmp(rc_pairs.data(), want_rc.data(), 2 * n * 8) == 0);

  // SegmentingWatershed::transform: Options::ffi() + ws_segment into a zeroed Array2<usize>
  ws_options o{};
  o.max_water_level = 254; o.edge_correction = 0; o.engine = 0; o.tie_rule = 0; o.seed_shift = 0;
  std::vector<uint64_t> out(H * W, 0), want(H * W);
  CHECK(ws_segment(ctx, img.data(), H, W, W, rc_pairs.data(), n, &o, out.data()) == WS_OK);
  CHECK(ws_or_segment_arrival(img.data(), H, W, rc_pairs.data(), n, 254, 0, want.data(), nullptr) == 0);
  CHECK(out == want);

  // Watershed::transform_history (run_with_hook + history_hook + shim::trampoline): per level a copy of the plane;
  // edge correction on, so planes are (H + 2) x (W + 2) and seeds index the padded plane unshifted (lib.rs:1675-1677)
  struct Hist { std::vector<std::vector<uint64_t>> planes; std::vector<uint8_t> levels; size_t h = 0, w = 0; } got_h, want_h;
  auto keep = [](void *u, uint8_t lvl, uint8_t, const uint8_t *, const uint64_t *lab, size_t h, size_t w) {
    Hist *hs = static_cast<Hist *>(u);
    hs->planes.emplace_back(lab, lab + h * w);
    hs->levels.push_back(lvl);
    hs->h = h; hs->w = w;
  };
  o.max_water_level = 37; o.edge_correction = 1;
  CHECK(ws_segment_with_hook(ctx, img.data(), H, W, W, rc_pairs.data(), n, &o, keep, &got_h, nullptr) == WS_OK);
  std::vector<uint64_t> scratch((H + 2) * (W + 2));
  CHECK(ws_or_segment(img.data(), H, W, rc_pairs.data(), n, 37, 1, WS_OR_TIE_FIRST, 0, scratch.data(), nullptr, nullptr, keep, &want_h, nullptr) == 0);
  CHECK(got_h.planes.size() == 38 && got_h.h == H + 2 && got_h.w == W + 2 && got_h.levels == want_h.levels);
  CHECK(got_h.planes == want_h.planes);

  // the same with ws_options.seed_shift = 1 (TransformBuilder::shift_seeds_into_padded_plane): equal to the
  // reference behaviour on seeds moved by (+1, +1)
  std::vector<uint64_t> moved(rc_pairs.begin(), rc_pairs.begin() + 2 * n);
  for (uint64_t &v : moved) v += 1;
  o.max_water_level = 254; o.seed_shift = 1;
  std::vector<uint64_t> pout((H + 2) * (W + 2)), pwant((H + 2) * (W + 2));
  CHECK(ws_segment(ctx, img.data(), H, W, W, rc_pairs.data(), n, &o, pout.data()) == WS_OK);
  CHECK(ws_or_segment_arrival(img.data(), H, W, moved.data(), n, 254, 1, pwant.data(), nullptr) == 0);
  CHECK(pout == pwant);
  o.seed_shift = 0;

  // Watershed::transform_to_list (run_to_list): first guess too small on purpose -> WS_ERR_CAPACITY with the exact
  // count -> second call; then the dense expansion to Vec<usize> of length pixels + 1
  o.max_water_level = 50; o.edge_correction = 0;
  std::vector<uint64_t> offsets(52), uncoloured(51);
  size_t n_lakes = 0, lcap = 16;
  std::vector<ws_lake> lakes(lcap);
  int rc = ws_transform_to_list(ctx, 1, img.data(), H, W, W, rc_pairs.data(), n, &o, lakes.data(), lcap, &n_lakes, offsets.data(), uncoloured.data());
  CHECK(rc == WS_ERR_CAPACITY && n_lakes > lcap);
  lcap = n_lakes;
  lakes.assign(lcap, ws_lake{});
  CHECK(ws_transform_to_list(ctx, 1, img.data(), H, W, W, rc_pairs.data(), n, &o, lakes.data(), lcap, &n_lakes, offsets.data(), uncoloured.data()) == WS_OK);
  struct Lists { std::vector<std::vector<uint64_t>> v; } want_l;
  auto sizes = [](void *u, uint8_t, uint8_t, const uint8_t *, const uint64_t *lab, size_t h, size_t w) {
    std::vector<uint64_t> hist(h * w + 1);
    ws_or_find_lake_sizes(lab, h * w, hist.data());
    static_cast<Lists *>(u)->v.push_back(std::move(hist));
  };
  CHECK(ws_or_merge_arrival(img.data(), H, W, rc_pairs.data(), n, 50, 0, want.data(), sizes, &want_l) == 0);
  CHECK(want_l.v.size() == 51);
  for (size_t l = 0; l < 51; ++l) {
    std::vector<uint64_t> dense(H * W + 1, 0);
    dense[0] = uncoloured[l];
    for (uint64_t k = offsets[l]; k < offsets[l + 1]; ++k) dense[lakes[k].colour] = lakes[k].area;
    CHECK(dense == want_l.v[l]);
  }

  // MergingWatershed::transform_final: ws_merge_with_hook, no callback, final labels out
  CHECK(ws_merge_with_hook(ctx, img.data(), H, W, W, rc_pairs.data(), n, &o, nullptr, nullptr, out.data()) == WS_OK);
  CHECK(out == want);
  // MergingWatershed::transform: the reference's stub
  CHECK(ws_merge_transform_stub(H, W, out.data()) == WS_OK);
  ws_or_merge_transform_stub(H, W, want.data());
  CHECK(out == want);

  // WatershedUtils::pre_processor_with_max::<MAX>: f32 goes through as WS_F32, an unlisted type as f64
  std::vector<float> f(H * W);
  for (size_t i = 0; i < f.size(); ++i) f[i] = (float)((int)(i % 977) - 300) * 0.37f;
  f[5] = 0.0f; f[6] = std::numeric_limits<float>::infinity(); f[7] = -std::numeric_limits<float>::infinity();
  f[8] = std::numeric_limits<float>::quiet_NaN();
  std::vector<uint8_t> q(f.size()), qw(f.size());
  CHECK(ws_pre_processor(ctx, f.data(), WS_F32, f.size(), 200, q.data()) == WS_OK);
  CHECK(ws_or_pre_processor(f.data(), 0, f.size(), 200, qw.data()) == 0);
  CHECK(q == qw);

  // shim::check: WS_ERR_SEED_OOB becomes the reference's index panic (lib.rs:1676)
  const uint64_t bad[2] = {H, 0};
  o.max_water_level = 254;
  CHECK(ws_segment(ctx, img.data(), H, W, W, bad, 1, &o, out.data()) == WS_ERR_SEED_OOB);
  CHECK(std::strlen(ws_last_error(ctx)) > 0);
  // HipCtx::drop
  ws_ctx_destroy(ctx);
  std::printf("shim sequence ok\n");
  return 0;
}

// Several ranks behind the same call (include/ws_hip.h, ws_group_*): what a TiledWatershed wrapper of the shim would do for
// transform() -- one local group, ws_segment_tiled with host buffers -- against the oracle's single-domain transform
// (lib.rs:1638-1808 and 1328-1522 in one address space).  Three ranks on device 0: the protocol, not the speed.
static int group_sequence_checks() {
  const size_t H = 300, W = 256;
  std::vector<uint8_t> img(H * W);
  ws_or_random_field(img.data(), H, W, 23);
  std::vector<uint64_t> rc_pairs(H * W);
  const size_t n = ws_or_find_local_minima(img.data(), H, W, rc_pairs.data(), H * W / 2);
  CHECK(n > 100);
  ws_group *g = nullptr;
  const int devices[3] = {0, 0, 0};
  CHECK(ws_group_create_local(3, devices, &g) == WS_OK);
  int world = 0, n_local = 0, first = -1;
  CHECK(ws_group_info(g, &world, &n_local, &first) == WS_OK && world == 3 && n_local == 3 && first == 0);
  CHECK(ws_group_selftest(g) == WS_OK);
  size_t r0, r1, lo, hi;
  CHECK(ws_tile_rows(H, 1, 3, &r0, &r1, &lo, &hi) == WS_OK && r0 == 100 && r1 == 200 && lo == 99 && hi == 201);
  ws_options o;
  ws_options_default(&o);
  std::vector<uint64_t> out(H * W), want(H * W);
  uint32_t rounds = 0;
  CHECK(ws_segment_tiled(g, img.data(), H, W, W, rc_pairs.data(), n, &o, 0, out.data(), &rounds) == WS_OK);
  CHECK(ws_or_segment_arrival(img.data(), H, W, rc_pairs.data(), n, 254, 0, want.data(), nullptr) == 0);
  CHECK(out == want && rounds >= 3);
  o.max_water_level = 90;
  CHECK(ws_segment_tiled(g, img.data(), H, W, W, rc_pairs.data(), n, &o, 1, out.data(), nullptr) == WS_OK);
  CHECK(ws_or_merge_arrival(img.data(), H, W, rc_pairs.data(), n, 90, 0, want.data(), nullptr, nullptr) == 0);
  CHECK(out == want);
  // the same field in 2 x 2 tiles (halo rows and columns) on a group of four, plain and with edge correction
  {
    ws_group *g4 = nullptr;
    const int dev4[4] = {0, 0, 0, 0};
    CHECK(ws_group_create_local(4, dev4, &g4) == WS_OK);
    size_t rows[4], cols[4];
    CHECK(ws_tile_grid(H, W, 3, 2, 2, rows, cols) == WS_OK && rows[0] == 150 && rows[2] == 149 && cols[0] == 128 && cols[2] == 127 && cols[3] == 256);
    ws_options o2;
    ws_options_default(&o2);
    uint32_t r2 = 0;
    CHECK(ws_segment_tiled2d(g4, img.data(), H, W, W, rc_pairs.data(), n, &o2, 2, 2, 0, out.data(), &r2) == WS_OK);
    CHECK(ws_or_segment_arrival(img.data(), H, W, rc_pairs.data(), n, 254, 0, want.data(), nullptr) == 0);
    CHECK(out == want && r2 >= 2);
    o2.edge_correction = 1;
    std::vector<uint64_t> oute((H + 2) * (W + 2)), wante((H + 2) * (W + 2));
    CHECK(ws_segment_tiled2d(g4, img.data(), H, W, W, rc_pairs.data(), n, &o2, 2, 2, 0, oute.data(), nullptr) == WS_OK);
    CHECK(ws_or_segment_arrival(img.data(), H, W, rc_pairs.data(), n, 254, 1, wante.data(), nullptr) == 0);
    CHECK(oute == wante);
    // the merging transform's final labels in tiles, at a level where lakes are not trivial
    o2.max_water_level = 90;
    CHECK(ws_segment_tiled2d(g4, img.data(), H, W, W, rc_pairs.data(), n, &o2, 2, 2, 1, oute.data(), nullptr) == WS_OK);
    CHECK(ws_or_merge_arrival(img.data(), H, W, rc_pairs.data(), n, 90, 1, wante.data(), nullptr, nullptr) == 0);
    CHECK(oute == wante);
    CHECK(ws_segment_tiled2d(g4, img.data(), H, W, W, rc_pairs.data(), n, &o2, 3, 2, 0, oute.data(), nullptr) == WS_ERR_BAD_ARG);
    ws_group_destroy(g4);
  }
  // transform_to_list of the field over the group against the one-context call: the same lakes and areas at every level
  {
    ws_options o3;
    ws_options_default(&o3);
    const size_t cap = 255 * (n + 1);
    std::vector<ws_lake> la(cap), lb(cap);
    std::vector<uint64_t> oa(256), ob(256), ua(255), ub(255);
    size_t na = 0, nb = 0;
    ws_ctx *c1 = nullptr;
    CHECK(ws_ctx_create(0, &c1) == WS_OK);
    CHECK(ws_transform_to_list(c1, 1, img.data(), H, W, W, rc_pairs.data(), n, &o3, la.data(), cap, &na, oa.data(), ua.data()) == WS_OK);
    CHECK(ws_transform_to_list_tiled(g, 1, img.data(), H, W, W, rc_pairs.data(), n, &o3, lb.data(), cap, &nb, ob.data(), ub.data(), nullptr) == WS_OK);
    CHECK(na == nb && oa == ob && ua == ub);
    auto by_colour = [](const ws_lake &x, const ws_lake &y) { return x.colour < y.colour; };
    for (size_t l = 0; l < 255; ++l) {      // (a level's records come in the order their waves wrote them)
      std::sort(la.begin() + oa[l], la.begin() + oa[l + 1], by_colour);
      std::sort(lb.begin() + ob[l], lb.begin() + ob[l + 1], by_colour);
    }
    bool same = true;
    for (size_t i = 0; i < na; ++i) same = same && la[i].colour == lb[i].colour && la[i].area == lb[i].area;
    CHECK(same);
    ws_ctx_destroy(c1);
  }
  const uint64_t bad[2] = {H, 0};
  CHECK(ws_segment_tiled(g, img.data(), H, W, W, bad, 1, &o, 0, out.data(), nullptr) == WS_ERR_SEED_OOB);
  CHECK(std::strlen(ws_group_last_error(g)) > 0);
  ws_group_destroy(g);
  std::printf("group sequence ok\n");
  return 0;
}

int main(int argc, char **argv) {
  const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
  if (cpu_checks()) return 1;
  if (gpu) return gpu_checks() || shim_sequence_checks() || group_sequence_checks();
  return no_device_check();
}
