// ws_watershed.hpp -- header-only C++ mirror of the reference's public interface for the
// watershed path, over the C ABI in ws_hip.h.
//
// The reference (smups/rustronomy-watershed v0.4.1) is Rust and this image has no Rust
// toolchain, so the host side above the C ABI is written in C++ with the reference's own
// names, argument meaning and error behaviour ("lib.rs:N" = src/lib.rs line N):
//
//   TransformBuilder<T>      lib.rs:908-1047      BuildErr               lib.rs:1051-1065
//   HookCtx                  lib.rs:844-862       WatershedUtils         lib.rs:1069-1198
//   Watershed<T> (4 methods) lib.rs:1206-1238     Segmenting/MergingWatershed  lib.rs:1297-1849
//
// Rust panics (out-of-bounds seed, lib.rs:1366/1676) surface as std::out_of_range; Result<_,
// BuildErr> as a thrown BuildErr.  Documented deviations from the reference: see ws_hip.h.
#pragma once

#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "ws_hip.h"

namespace rustronomy_watershed {

constexpr std::size_t UNCOLOURED = 0;      // lib.rs:138
constexpr std::uint8_t NORMAL_MAX = 254;   // lib.rs:139
constexpr std::uint8_t ALWAYS_FILL = 0;    // lib.rs:140
constexpr std::uint8_t NEVER_FILL = 255;   // lib.rs:141

using usize = std::uint64_t;
using Seed = std::pair<usize, usize>;      // (row, col)

struct BuildErr : std::runtime_error {     // lib.rs:1051-1065
  enum Kind { MaxToHigh, MaxToLow } kind;
  std::uint8_t value;
  BuildErr(Kind k, std::uint8_t v)
      : std::runtime_error(k == MaxToHigh
                               ? "Maximum water level set to " + std::to_string(v) + ", which is higher than the maximum allowed value 254"
                               : "Maximum water level set to " + std::to_string(v) + ", which is lower than the minimum allowed value 255"),
        kind(k), value(v) {}
};

struct WatershedError : std::runtime_error {
  int status;
  WatershedError(int s, const std::string &detail) : std::runtime_error(std::string(ws_strerror(s)) + ": " + detail), status(s) {}
};

template <class T>
struct Array2 {                            // owned, standard layout (ndarray::Array2<T>)
  std::size_t rows = 0, cols = 0;
  std::vector<T> data;
  Array2() = default;
  Array2(std::size_t r, std::size_t c) : rows(r), cols(c), data(r * c) {}
  T &operator()(std::size_t r, std::size_t c) { return data[r * cols + c]; }
  const T &operator()(std::size_t r, std::size_t c) const { return data[r * cols + c]; }
};

template <class T>
struct ArrayView2 {                        // borrowed view with a row stride in elements
  const T *ptr = nullptr;
  std::size_t rows = 0, cols = 0, row_stride = 0;
  ArrayView2() = default;
  ArrayView2(const T *p, std::size_t r, std::size_t c, std::size_t s) : ptr(p), rows(r), cols(c), row_stride(s) {}
  ArrayView2(const Array2<T> &a) : ptr(a.data.data()), rows(a.rows), cols(a.cols), row_stride(a.cols) {}
  const T &operator()(std::size_t r, std::size_t c) const { return ptr[r * row_stride + c]; }
};

struct HookCtx {                           // lib.rs:844-862
  std::uint8_t water_level, max_water_level;
  ArrayView2<std::uint8_t> image;
  ArrayView2<usize> colours;
  const std::vector<std::pair<usize, Seed>> *seeds;    // (colour, (row, col)), lib.rs:1671-1672
};

// One ws_ctx per object; not thread safe (make one per thread: the reference's structs are Send + Sync).
class Context {
 public:
  explicit Context(int device = 0) {
    int rc = ws_ctx_create(device, &ctx_);
    if (rc != WS_OK) throw WatershedError(rc, "ws_ctx_create");
  }
  ~Context() { ws_ctx_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  ws_ctx *get() const { return ctx_; }
  void check(int rc) const {
    if (rc == WS_OK) return;
    if (rc == WS_ERR_SEED_OOB) throw std::out_of_range(ws_last_error(ctx_));
    throw WatershedError(rc, ws_last_error(ctx_));
  }

 private:
  ws_ctx *ctx_ = nullptr;
};

namespace detail {
template <class T>
struct HookBox {
  std::function<T(const HookCtx &)> fn;
  std::vector<T> out;
  const std::vector<std::pair<usize, Seed>> *seeds;
  static void thunk(void *user, std::uint8_t lvl, std::uint8_t mx, const std::uint8_t *img, const std::uint64_t *lab,
                    std::size_t h, std::size_t w) {
    auto *self = static_cast<HookBox *>(user);
    HookCtx ctx{lvl, mx, ArrayView2<std::uint8_t>(img, h, w, w), ArrayView2<usize>(lab, h, w, w), self->seeds};
    self->out.push_back(self->fn(ctx));
  }
};
inline std::vector<std::uint64_t> pack(const std::vector<Seed> &seeds) {
  std::vector<std::uint64_t> p(2 * seeds.size());
  for (std::size_t i = 0; i < seeds.size(); ++i) { p[2 * i] = seeds[i].first; p[2 * i + 1] = seeds[i].second; }
  return p;
}
}  // namespace detail

template <class E> struct dtype_of;
template <> struct dtype_of<float> { static constexpr int value = WS_F32; };
template <> struct dtype_of<double> { static constexpr int value = WS_F64; };
template <> struct dtype_of<std::int32_t> { static constexpr int value = WS_I32; };
template <> struct dtype_of<std::uint16_t> { static constexpr int value = WS_U16; };
template <> struct dtype_of<std::int16_t> { static constexpr int value = WS_I16; };
template <> struct dtype_of<std::uint8_t> { static constexpr int value = WS_U8; };

class WatershedUtils {                     // lib.rs:1069-1198
 public:
  // lib.rs:1134-1173: pre_processor_with_max::<MAX, T, D>; any dimension, passed as a flat span
  template <std::uint8_t MAX, class E>
  std::vector<std::uint8_t> pre_processor_with_max(const E *data, std::size_t n) const {
    static_assert(MAX < NEVER_FILL && MAX > ALWAYS_FILL, "lib.rs:1143-1144");
    std::vector<std::uint8_t> out(n);
    ctx_->check(ws_pre_processor(ctx_->get(), data, dtype_of<E>::value, n, MAX, out.data()));
    return out;
  }
  template <class E>
  std::vector<std::uint8_t> pre_processor(const E *data, std::size_t n) const {   // lib.rs:1081-1087
    return pre_processor_with_max<NORMAL_MAX, E>(data, n);
  }

  std::vector<Seed> find_local_minima(ArrayView2<std::uint8_t> img) const {
    const std::size_t cap = ((img.rows ? img.rows - 1 : 0) / 2 + 1) * ((img.cols ? img.cols - 1 : 0) / 2 + 1);
    std::vector<std::uint64_t> rc(2 * cap);
    std::size_t n = 0;
    ctx_->check(ws_find_local_minima(ctx_->get(), img.ptr, img.rows, img.cols, img.row_stride, rc.data(), cap, &n));
    std::vector<Seed> out(n);
    for (std::size_t i = 0; i < n; ++i) out[i] = {rc[2 * i], rc[2 * i + 1]};
    return out;
  }

 protected:
  explicit WatershedUtils(std::shared_ptr<Context> c) : ctx_(std::move(c)) {}
  std::shared_ptr<Context> ctx_;
};

template <class T>
class Watershed : public WatershedUtils {  // lib.rs:1206-1238
 public:
  using Hook = std::function<T(const HookCtx &)>;
  std::uint8_t max_water_level() const { return opt_.max_water_level; }
  bool edge_correction() const { return opt_.edge_correction != 0; }

  std::vector<T> transform_with_hook(ArrayView2<std::uint8_t> input, const std::vector<Seed> &seeds) const {
    if (!hook_) { run<int>(input, seeds, nullptr, nullptr); return {}; }   // lib.rs:1796-1807: work done, empty Vec
    return run<T>(input, seeds, &hook_, nullptr);
  }
  std::vector<std::pair<std::uint8_t, Array2<usize>>> transform_history(ArrayView2<std::uint8_t> input,
                                                                          const std::vector<Seed> &seeds) const {
    using R = std::pair<std::uint8_t, Array2<usize>>;
    std::function<R(const HookCtx &)> h = [](const HookCtx &c) {           // lib.rs:1545 / 1831
      Array2<usize> a(c.colours.rows, c.colours.cols);
      for (std::size_t r = 0; r < a.rows; ++r)
        for (std::size_t q = 0; q < a.cols; ++q) a(r, q) = c.colours(r, q);
      return R{c.water_level, std::move(a)};
    };
    return run<R>(input, seeds, &h, nullptr);
  }
  // lib.rs:1220-1224: Vec<(u8, Vec<usize>)>, each inner Vec of length pixels+1 (lib.rs:630)
  std::vector<std::pair<std::uint8_t, std::vector<usize>>> transform_to_list(ArrayView2<std::uint8_t> input,
                                                                              const std::vector<Seed> &seeds) const {
    const std::size_t e = opt_.edge_correction ? 2 : 0, npx = (input.rows + e) * (input.cols + e);
    const std::size_t levels = std::size_t(opt_.max_water_level) + 1;
    std::vector<std::uint64_t> offsets(levels + 1), unc(levels);
    auto packed = detail::pack(seeds);
    // one record per live lake and level, at most seeds * levels; half of that covers a random field.  A too
    // small guess costs a second transform, not a wrong answer.
    std::size_t cap = std::min<std::size_t>(seeds.size() * levels / 2 + 1024, std::size_t(1) << 26), n = 0;
    std::vector<ws_lake> lakes;
    for (;;) {
      lakes.resize(cap);
      int rc = ws_transform_to_list(ctx_->get(), merging_, input.ptr, input.rows, input.cols, input.row_stride, packed.data(),
                                    seeds.size(), &opt_, lakes.data(), cap, &n, offsets.data(), unc.data());
      if (rc == WS_ERR_CAPACITY && n > cap) { cap = n; continue; }
      ctx_->check(rc);
      break;
    }
    std::vector<std::pair<std::uint8_t, std::vector<usize>>> out;
    for (std::size_t l = 0; l < levels; ++l) {
      std::vector<usize> hist(npx + 1, 0);
      for (std::uint64_t i = offsets[l]; i < offsets[l + 1]; ++i) hist[lakes[i].colour] = lakes[i].area;
      hist[0] = unc[l];
      out.emplace_back(std::uint8_t(l), std::move(hist));
    }
    return out;
  }

 protected:
  Watershed(ws_options o, Hook h, std::shared_ptr<Context> c, int merging)
      : WatershedUtils(std::move(c)), opt_(o), hook_(std::move(h)), merging_(merging) {}

  template <class R>
  std::vector<R> run(ArrayView2<std::uint8_t> input, const std::vector<Seed> &seeds, const std::function<R(const HookCtx &)> *hook,
                     Array2<usize> *final_labels) const {
    auto packed = detail::pack(seeds);
    std::vector<std::pair<usize, Seed>> seed_colours;
    detail::HookBox<R> box;
    if (hook) {
      for (std::size_t i = 0; i < seeds.size(); ++i) seed_colours.push_back({i + 1, seeds[i]});
      box.fn = *hook;
      box.seeds = &seed_colours;
    }
    auto fn = merging_ ? ws_merge_with_hook : ws_segment_with_hook;
    ctx_->check(fn(ctx_->get(), input.ptr, input.rows, input.cols, input.row_stride, packed.data(), seeds.size(), &opt_,
                   hook ? &detail::HookBox<R>::thunk : nullptr, hook ? &box : nullptr,
                   final_labels ? final_labels->data.data() : nullptr));
    return std::move(box.out);
  }

  ws_options opt_;
  Hook hook_;
  int merging_;
};

template <class T = int>
class SegmentingWatershed : public Watershed<T> {     // lib.rs:1609-1849
 public:
  // lib.rs:1810-1822, intended semantics (the reference's body panics): labels after the last level
  Array2<usize> transform(ArrayView2<std::uint8_t> input, const std::vector<Seed> &seeds) const {
    const std::size_t e = this->opt_.edge_correction ? 2 : 0;
    Array2<usize> out(input.rows + e, input.cols + e);
    this->template run<int>(input, seeds, nullptr, &out);
    return out;
  }
  // Not in the reference: its README's call pair, transform(input, find_local_minima(input)) (lib.rs:73-86), as ONE call of the
  // library (ws_segment_minima).  `seeds_out` (optional) receives the list the labels are numbered by.
  Array2<usize> transform_from_minima(ArrayView2<std::uint8_t> input, std::vector<Seed> *seeds_out = nullptr) const {
    const std::size_t e = this->opt_.edge_correction ? 2 : 0;
    Array2<usize> out(input.rows + e, input.cols + e);
    const std::size_t cap = seeds_out && input.rows >= 3 && input.cols >= 3 ? ((input.rows - 1) / 2 + 1) * ((input.cols - 1) / 2 + 1) : 0;
    std::vector<std::uint64_t> rc(2 * cap + 2);
    std::size_t n = 0;
    this->ctx_->check(ws_segment_minima(this->ctx_->get(), input.ptr, input.rows, input.cols, input.row_stride, &this->opt_,
                                        reinterpret_cast<std::uint64_t *>(out.data.data()), cap ? rc.data() : nullptr, cap, &n));
    if (seeds_out) {
      seeds_out->resize(n);
      for (std::size_t i = 0; i < n; ++i) (*seeds_out)[i] = {(usize)rc[2 * i], (usize)rc[2 * i + 1]};
    }
    return out;
  }
  // Not in the reference: what its integration tests loop over (tests/integration.rs:267,356 -- find_local_minima + transform for
  // one slice of a cube after the other) as ONE call of the library (ws_segment_batch: the slices' uploads, transforms and label
  // copies overlap).  `cube`: n_slices contiguous rows x cols slices; slice k of the result is transform_from_minima of slice k.
  std::vector<Array2<usize>> transform_cube(const std::uint8_t *cube, std::size_t n_slices, std::size_t rows, std::size_t cols) const {
    const std::size_t e = this->opt_.edge_correction ? 2 : 0, plane = (rows + e) * (cols + e);
    std::vector<std::uint64_t> flat(n_slices * plane + 1);
    std::size_t failed = 0;
    this->ctx_->check(ws_segment_batch(this->ctx_->get(), cube, n_slices, rows, cols, cols, rows * cols, nullptr, nullptr, &this->opt_, flat.data(),
                                       nullptr, &failed));
    std::vector<Array2<usize>> out(n_slices, Array2<usize>(rows + e, cols + e));
    for (std::size_t k = 0; k < n_slices; ++k)
      for (std::size_t i = 0; i < plane; ++i) out[k].data[i] = (usize)flat[k * plane + i];
    return out;
  }

 private:
  template <class U> friend class TransformBuilder;
  using Watershed<T>::Watershed;
};

template <class T = int>
class MergingWatershed : public Watershed<T> {        // lib.rs:1297-1562
 public:
  // lib.rs:1524-1536: a stub in the reference (zeros, interior 123, seeds ignored)
  Array2<usize> transform(ArrayView2<std::uint8_t> input, const std::vector<Seed> &) const {
    Array2<usize> out(input.rows, input.cols);
    int rc = ws_merge_transform_stub(input.rows, input.cols, out.data.data());
    if (rc != WS_OK) throw WatershedError(rc, "ws_merge_transform_stub");
    return out;
  }
  // not in the reference: merged labels after the last level (canonical ids)
  Array2<usize> transform_final(ArrayView2<std::uint8_t> input, const std::vector<Seed> &seeds) const {
    const std::size_t e = this->opt_.edge_correction ? 2 : 0;
    Array2<usize> out(input.rows + e, input.cols + e);
    this->template run<int>(input, seeds, nullptr, &out);
    return out;
  }

 private:
  template <class U> friend class TransformBuilder;
  using Watershed<T>::Watershed;
};

template <class T = int>
class TransformBuilder {                   // lib.rs:908-1047
 public:
  TransformBuilder() { ws_options_default(&opt_); }                                  // lib.rs:936-946
  static TransformBuilder new_() { return TransformBuilder(); }
  TransformBuilder &set_max_water_lvl(std::uint8_t v) { opt_.max_water_level = v; return *this; }   // lib.rs:950
  TransformBuilder &enable_edge_correction() { opt_.edge_correction = 1; return *this; }             // lib.rs:958
  TransformBuilder &set_wlvl_hook(std::function<T(const HookCtx &)> h) { hook_ = std::move(h); return *this; }   // lib.rs:967
  TransformBuilder &set_engine(ws_engine e) { opt_.engine = std::uint8_t(e); return *this; }        // this implementation only
  TransformBuilder &set_context(std::shared_ptr<Context> c) { ctx_ = std::move(c); return *this; }
  // this implementation only (ws_options.seed_shift): with edge correction, seeds move by (+1, +1) onto their own pixel
  // instead of indexing the padded plane with the caller's coordinates (lib.rs:1675-1677)
  TransformBuilder &shift_seeds_into_padded_plane(bool on = true) { opt_.seed_shift = on ? 1 : 0; return *this; }

  SegmentingWatershed<T> build_segmenting() const {                                                  // lib.rs:1024-1046
    validate();
    return SegmentingWatershed<T>(opt_, hook_, ctx_ ? ctx_ : std::make_shared<Context>(0), 0);
  }
  MergingWatershed<T> build_merging() const {                                                        // lib.rs:998-1020
    validate();
    return MergingWatershed<T>(opt_, hook_, ctx_ ? ctx_ : std::make_shared<Context>(0), 1);
  }

 private:
  void validate() const {
    const int rc = ws_options_validate(&opt_);
    if (rc == WS_ERR_MAX_TOO_HIGH) throw BuildErr(BuildErr::MaxToHigh, opt_.max_water_level);     // lib.rs:1026-1027
    if (rc == WS_ERR_MAX_TOO_LOW) throw BuildErr(BuildErr::MaxToLow, opt_.max_water_level);       // lib.rs:1028-1029
    if (rc != WS_OK) throw WatershedError(rc, "ws_options_validate");
  }
  ws_options opt_;
  std::function<T(const HookCtx &)> hook_;
  std::shared_ptr<Context> ctx_;
};

}  // namespace rustronomy_watershed
