// ws_ctx.hip -- context life cycle, options and errors, the seed finder, the pre-processor, staging of host inputs.
//
// There is no CPU fallback: every entry point that computes needs a HIP device and fails with WS_ERR_NO_DEVICE /
// WS_ERR_HIP otherwise.
#include "ws_ctx.hpp"

namespace wsapi {

PassFlags make_pf(ws_ctx *c) {
  uint32_t *f = (uint32_t *)c->flags.p;
  c->misc_clean = false;      // the overflow word may be written
  return PassFlags{f + FLAG_EDGE, f + FLAG_ANY, f + FLAG_OVERFLOW, c->profiling ? f + FLAG_STATS : nullptr};
}

bool slot_nonzero(const uint32_t *slot) {
  uint32_t any = 0;
  for (int i = 0; i < NSTRIPE; ++i) any |= slot[i * STRIPE_STRIDE];
  return any != 0;
}

int fail(ws_ctx *c, int code, const char *what, hipError_t e) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

int ensure(ws_ctx *c, DevBuf &b, size_t bytes) {
  if (bytes <= b.cap) return WS_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  ++c->buffer_generation;      // a captured graph holds the old pointer
  const size_t want = bytes + (bytes >> 3) + 256;     // a little slack so near-equal sizes reuse
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; return fail(c, WS_ERR_OOM, "hipMalloc", e); }
  b.cap = want;
  return WS_OK;
}

hipEvent_t next_event(ws_ctx *c) {
  if (c->ev_used == c->ev_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    c->ev_pool.push_back(e);
  }
  return c->ev_pool[c->ev_used++];
}

void stats_begin(ws_ctx *c) {
  std::memset(&c->stats, 0, sizeof c->stats);
  c->spans.clear();
  c->ev_used = 0;
  (void)hipEventRecord(c->ev_begin, c->stream);
}

int stats_end(ws_ctx *c) {
  // (the second half of a begun transform has waited for ITS work already; the stream may hold the next context's)
  if (c->stats_no_wait) { c->stats.ms_total = 0.0f; return WS_OK; }
  HIP_TRY(c, hipEventRecord(c->ev_end, c->stream));
  HIP_TRY(c, hipEventSynchronize(c->ev_end));
  float ms = 0;
  if (hipEventElapsedTime(&ms, c->ev_begin, c->ev_end) == hipSuccess) c->stats.ms_total = ms;
  for (const TimedSpan &s : c->spans) {
    float t = 0;
    if (hipEventElapsedTime(&t, s.a, s.b) != hipSuccess) continue;
    switch (s.cls) {
      case KC_RELAX: c->stats.ms_relax += t; break;
      case KC_RESOLVE: c->stats.ms_resolve += t; break;
      case KC_SWEEP: c->stats.ms_sweep += t; break;
      default: c->stats.ms_other += t; break;
    }
  }
  return WS_OK;
}

int check_plane(ws_ctx *c, size_t h, size_t w, size_t stride, const ws_options *opt, size_t *ph, size_t *pw) {
  if (!opt) return fail(c, WS_ERR_BAD_ARG, "options pointer is null");
  int v = ws_options_validate(opt);
  if (v != WS_OK) return fail(c, v, ws_strerror(v));
  if (stride < w) return fail(c, WS_ERR_BAD_ARG, "row_stride < w");
  const size_t e = opt->edge_correction ? 2 : 0;
  *ph = h + e;
  *pw = w + e;
  if (*ph > 0x7FFFFFF0ull || *pw > 0x7FFFFFF0ull || (*ph) * (*pw) >= 0xFFFFFFFFull)
    return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^32 pixels");
  return WS_OK;
}

// With edge correction the kernels read the caller's own image through a virtual ring of zeros (padded_img_index):
// no padded copy exists.  An EMPTY image (h == 0 or w == 0) still has a plane of border pixels, never flooded, and the
// kernels read a clamped address for border pixels, which must exist: padded_img_index then spans the long side of the
// plane (column index up to w - 1 when h == 0, row index up to h - 1 when w == 0), so a zeroed block of max(h, w) + 16
// bytes of the context stands in, with a row stride of 1.
int empty_image_block(ws_ctx *c, const uint8_t **d_img, size_t *d_stride, size_t h, size_t w) {
  int rc;
  const size_t bytes = std::max(h, w) + 16;
  if ((rc = ensure(c, c->img, bytes))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->img.p, 0, bytes, c->stream));
  *d_img = (const uint8_t *)c->img.p;
  *d_stride = 1;
  return WS_OK;
}

// Uploads a host image and host seeds; returns device pointers (the image tightly packed, stride w).
int stage_inputs(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                 size_t n_seeds, const ws_options *opt, size_t ph, size_t pw, const uint8_t **d_img,
                 size_t *d_stride, const uint32_t **d_seeds) {
  if ((!img && h * w) || (!seeds_rc && n_seeds)) return fail(c, WS_ERR_BAD_ARG, "null input pointer");
  if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
  int rc;
  // seeds: the reference indexes the (padded) plane with the caller's coordinates and panics
  // when they fall outside (lib.rs:1675-1677)
  // (checked and narrowed to 32 bits on the device: a seed outside the plane becomes ~0 and raises the seed-error
  // word of the transform that follows)
  if ((rc = ensure(c, c->seeds, (n_seeds ? n_seeds : 1) * 2 * sizeof(uint32_t)))) return rc;
  if (n_seeds) {
    if ((rc = ensure(c, c->seeds64, n_seeds * 2 * sizeof(uint64_t)))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->seeds64.p, seeds_rc, n_seeds * 2 * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, narrow_seeds(c->stream, (const uint64_t *)c->seeds64.p, n_seeds, ph, pw, (uint32_t *)c->seeds.p, seed_shift_of(opt)));
  }
  *d_seeds = (const uint32_t *)c->seeds.p;

  if (h * w == 0) return empty_image_block(c, d_img, d_stride, h, w);
  if ((rc = ensure(c, c->img, h * w))) return rc;
  if (stride == w) HIP_TRY(c, hipMemcpyAsync(c->img.p, img, h * w, hipMemcpyHostToDevice, c->stream));
  else HIP_TRY(c, hipMemcpy2DAsync(c->img.p, w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  *d_img = (const uint8_t *)c->img.p;
  *d_stride = w;
  return WS_OK;
}

// Device seed list moved by (+1, +1) into the context's own buffer (seed_shift with edge correction).
int shifted_seeds(ws_ctx *c, const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt, const uint32_t **out) {
  *out = d_seeds_rc;
  if (!seed_shift_of(opt) || n_seeds == 0) return WS_OK;
  int rc;
  if ((rc = ensure(c, c->seeds, n_seeds * 2 * sizeof(uint32_t)))) return rc;
  HIP_TRY(c, shift_seeds(c->stream, d_seeds_rc, n_seeds, 1u, (uint32_t *)c->seeds.p));
  *out = (const uint32_t *)c->seeds.p;
  return WS_OK;
}

// contiguous host copy of the (padded) image for the hook's `image` argument
const uint8_t *hook_image(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, bool edge) {
  if (!edge && stride == w) return img;
  const size_t ph = h + (edge ? 2 : 0), pw = w + (edge ? 2 : 0), o = edge ? 1 : 0;
  c->host_img.assign(ph * pw ? ph * pw : 1, 0);
  for (size_t r = 0; r < h; ++r) std::memcpy(&c->host_img[(r + o) * pw + o], img + r * stride, w);
  return c->host_img.data();
}

}  // namespace wsapi

using namespace wsapi;

extern "C" {

int ws_abi_version(void) { return WS_ABI_VERSION; }

const char *ws_strerror(int status) {
  switch (status) {
    case WS_OK: return "ok";
    case WS_ERR_BAD_ARG: return "bad argument";
    case WS_ERR_MAX_TOO_HIGH: return "maximum water level higher than the maximum allowed value 254";
    case WS_ERR_MAX_TOO_LOW: return "maximum water level lower than the minimum allowed value 1";
    case WS_ERR_SEED_OOB: return "seed outside the label plane";
    case WS_ERR_HIP: return "HIP runtime error";
    case WS_ERR_OOM: return "out of memory";
    case WS_ERR_NO_DEVICE: return "no HIP device";
    case WS_ERR_CAPACITY: return "output buffer too small";
    case WS_ERR_RING_OVERFLOW: return "ring counter overflow";
    case WS_ERR_TOO_LARGE: return "input too large";
    case WS_ERR_UNSUPPORTED: return "unsupported";
    case WS_ERR_RCCL: return "RCCL error";
    default: return "unknown status";
  }
}

int ws_options_default(ws_options *out) {
  if (!out) return WS_ERR_BAD_ARG;
  out->max_water_level = WS_NORMAL_MAX;   // lib.rs:942
  out->edge_correction = 0;               // lib.rs:943
  out->engine = WS_ENGINE_AUTO;
  out->tie_rule = WS_TIE_FIRST_DRLU;
  out->seed_shift = 0;                    // lib.rs:1675-1677: seeds are not moved into the padded plane
  out->reserved[0] = out->reserved[1] = out->reserved[2] = 0;
  return WS_OK;
}

int ws_options_validate(const ws_options *opt) {
  if (!opt) return WS_ERR_BAD_ARG;
  if (opt->max_water_level > WS_NORMAL_MAX) return WS_ERR_MAX_TOO_HIGH;     // lib.rs:1026-1027
  if (opt->max_water_level <= WS_ALWAYS_FILL) return WS_ERR_MAX_TOO_LOW;     // lib.rs:1028-1029
  if (opt->edge_correction > 1 || opt->engine > WS_ENGINE_SWEEP || opt->tie_rule != WS_TIE_FIRST_DRLU) return WS_ERR_BAD_ARG;
  if (opt->seed_shift > 1 || opt->reserved[0] || opt->reserved[1] || opt->reserved[2]) return WS_ERR_BAD_ARG;
  return WS_OK;
}

static int ctx_create(int device, void *stream, bool own, ws_ctx **out) {
  if (!out) return WS_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return WS_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return WS_ERR_BAD_ARG;
  ws_ctx *c = new (std::nothrow) ws_ctx();
  if (!c) return WS_ERR_OOM;
  c->device = device;
  bool ok = hipSetDevice(device) == hipSuccess;
  if (ok && own) ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
  if (ok && !own) c->stream = (hipStream_t)stream;
  c->own_stream = own;
  if (const char *e = tuning_env("WS_DEBUG_MAXIT")) c->debug_max_iters = (uint32_t)std::atoi(e);
  ok = ok && hipHostMalloc((void **)&c->pinned, FLAG_WORDS * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
  if (ok && hipHostGetDevicePointer((void **)&c->pinned_dev, c->pinned, 0) != hipSuccess) { (void)hipGetLastError(); c->pinned_dev = nullptr; }
  ok = ok && hipEventCreate(&c->ev_begin) == hipSuccess && hipEventCreate(&c->ev_end) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->async_ev, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < COUNTER_RING; ++i) ok = hipEventCreateWithFlags(&c->ring_ev[i], hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < COUNTER_RING; ++i) ok = hipEventCreateWithFlags(&c->kern_ev[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) == hipSuccess;
  ok = ok && ensure(c, c->flags, FLAG_WORDS * sizeof(uint32_t)) == WS_OK;
  if (!ok) { ws_ctx_destroy(c); return WS_ERR_HIP; }
  *out = c;
  return WS_OK;
}

int ws_ctx_create(int device, ws_ctx **out) { return ctx_create(device, nullptr, true, out); }
int ws_ctx_create_on_stream(int device, void *hip_stream, ws_ctx **out) { return ctx_create(device, hip_stream, false, out); }

void ws_ctx_destroy(ws_ctx *c) {
  if (!c) return;
  for (ws_ctx *lane : c->lanes) ws_ctx_destroy(lane);
  c->lanes.clear();
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (DevBuf *b : {&c->img, &c->keys, &c->labels, &c->labels2, &c->stamps, &c->flags, &c->seeds, &c->out64, &c->counts, &c->aux, &c->seed_stack, &c->min_counts, &c->min_nibbles, &c->seeds64,
                    &c->uf_parent, &c->uf_size, &c->uf_hooked, &c->uf_death, &c->uf_sd, &c->alive, &c->px_items, &c->edge_items, &c->mflags, &c->lakes, &c->refs, &c->seed_tab, &c->tile_list})
    if (b->p) (void)hipFree(b->p);
  if (c->pinned) (void)hipHostFree(c->pinned);
  host_copy_release(c);
  if (c->graph_exec) (void)hipGraphExecDestroy(c->graph_exec);
  for (hipGraphExec_t g : c->list_graphs) if (g) (void)hipGraphExecDestroy(g);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < COUNTER_RING; ++i) if (c->ring_ev[i]) (void)hipEventDestroy(c->ring_ev[i]);
  for (int i = 0; i < COUNTER_RING; ++i) if (c->kern_ev[i]) (void)hipEventDestroy(c->kern_ev[i]);
  if (c->copy_stream) { (void)hipStreamSynchronize(c->copy_stream); (void)hipStreamDestroy(c->copy_stream); }
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->async_ev) (void)hipEventDestroy(c->async_ev);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *ws_last_error(const ws_ctx *c) { return c ? c->err.c_str() : "null context"; }

int ws_ctx_set_profiling(ws_ctx *c, int enabled) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  c->profiling = enabled != 0;
  return WS_OK;
}

int ws_ctx_get_stats(const ws_ctx *c, ws_stats *out) {
  if (!c || !out) return WS_ERR_BAD_ARG;
  *out = c->stats;
  return WS_OK;
}

int ws_ctx_set_batch_pixel_limit(ws_ctx *c, size_t max_px) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  c->batch_max_px = max_px == 0 ? 0x7FFFFFFFull : std::min<size_t>(max_px, 0x7FFFFFFFull);
  return WS_OK;
}

int ws_ctx_set_seam_repair_min_pixels(ws_ctx *c, size_t min_px) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  c->seam_min_px = min_px == 0 ? (size_t)1 << 24 : min_px;
  ++c->buffer_generation;      // a captured graph holds the launches of the other flow
  return WS_OK;
}

int ws_ctx_set_persistent_pass(ws_ctx *c, int mode) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || mode < 0 || mode > 4) return WS_ERR_BAD_ARG;
  c->persistent_pass = mode;
  ++c->buffer_generation;      // a captured graph holds the launches of the other form
  return WS_OK;
}

int ws_ctx_set_host_threads(ws_ctx *c, int n_threads) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || n_threads < 0 || n_threads > 64) return WS_ERR_BAD_ARG;
  c->host_threads = n_threads;
  return WS_OK;
}

int ws_ctx_set_live_list_min_colours(ws_ctx *c, size_t min_colours) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  c->live_list_min = min_colours == 0 ? (size_t)1 << 20 : min_colours;
  ++c->buffer_generation;      // captured level graphs hold the launches of the other form
  return WS_OK;
}

int ws_ctx_synchronize(ws_ctx *c) {
  if (!c) return WS_ERR_BAD_ARG;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return WS_OK;
}

// ---- seeds --------------------------------------------------------------------------------

int ws_find_local_minima_device(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride,
                                uint32_t *d_out_rc, size_t cap, size_t *n_found) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !n_found || (!d_img && h * w) || (!d_out_rc && cap)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (stride < w) return fail(c, WS_ERR_BAD_ARG, "row_stride < w");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^32 pixels");
  *n_found = 0;
  if (h < 3 || w < 3) return WS_OK;                          // no 3x3 window (lib.rs:1183)
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t nseg = minima_segments((int)h, (int)w);
  int rc;
  if ((rc = ensure(c, c->counts, nseg * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->aux, minima_mask_bytes((int)h, (int)w)))) return rc;
  uint32_t *counts = (uint32_t *)c->counts.p;
  uint8_t *nibbles = (uint8_t *)c->aux.p;
  uint32_t *flags = (uint32_t *)c->flags.p;
  HIP_TRY(c, minima_count(c->stream, d_img, stride, (int)h, (int)w, counts, nibbles));
  HIP_TRY(c, minima_write(c->stream, nibbles, (int)h, (int)w, counts, flags + FLAG_TOTAL, d_out_rc, cap));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_TOTAL], flags + FLAG_TOTAL, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *n_found = c->pinned[FLAG_TOTAL];
  if (*n_found > cap) return fail(c, WS_ERR_CAPACITY, "seed buffer too small");
  return WS_OK;
}

int ws_find_local_minima(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, uint64_t *out_rc,
                         size_t cap, size_t *n_found) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !n_found || (!img && h * w) || (!out_rc && cap)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (stride < w) return fail(c, WS_ERR_BAD_ARG, "row_stride < w");
  *n_found = 0;
  if (h < 3 || w < 3) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->img, h * w))) return rc;
  // at most one strict maximum per 2x2 block of the interior
  const size_t bound = ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1);
  const size_t dcap = std::min(cap, bound);
  if ((rc = ensure(c, c->seeds, (dcap ? dcap : 1) * 2 * sizeof(uint32_t)))) return rc;
  HIP_TRY(c, hipMemcpy2DAsync(c->img.p, w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  rc = ws_find_local_minima_device(c, (const uint8_t *)c->img.p, h, w, w, (uint32_t *)c->seeds.p, dcap, n_found);
  if (rc != WS_OK && rc != WS_ERR_CAPACITY) return rc;
  const size_t got = std::min(*n_found, dcap);
  if (got) {      // (the pairs are 2 x got words: a long list crosses the bus as u32 and is widened by host threads, ws_hostcopy.hip)
    if (int rc_copy = labels_to_host_u64(c, (const uint32_t *)c->seeds.p, out_rc, got * 2)) return rc_copy;
  }
  return rc;
}

int ws_random_field_device(ws_ctx *c, uint8_t *d_img, size_t h, size_t w, size_t stride, uint64_t seed) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || (!d_img && h * w) || stride < w) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull) return fail(c, WS_ERR_TOO_LARGE, "too large");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, random_field(c->stream, d_img, stride, (int)h, (int)w, seed));
  return WS_OK;
}

int ws_merge_transform_stub(size_t h, size_t w, uint64_t *out) {
  if (!out && h * w) return WS_ERR_BAD_ARG;
  std::memset(out, 0, h * w * sizeof(uint64_t));                       // lib.rs:1529
  if (h < 2 || w < 2) return WS_OK;
  for (size_t r = 1; r + 1 < h; ++r)
    for (size_t col = 1; col + 1 < w; ++col) out[r * w + col] = 123;   // lib.rs:1532
  return WS_OK;
}

// ---- pre-processor (lib.rs:1081-1173) --------------------------------------------------------------

int ws_pre_processor_device(ws_ctx *c, const void *d_data, int dtype, size_t n, uint8_t max_value, uint8_t *d_out) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || (n && (!d_data || !d_out))) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (preproc_elem_size(dtype) == 0) return fail(c, WS_ERR_BAD_ARG, "unknown dtype");
  if (max_value >= WS_NEVER_FILL) return fail(c, WS_ERR_MAX_TOO_HIGH, "MAX must be < NEVER_FILL (lib.rs:1143)");
  if (max_value <= WS_ALWAYS_FILL) return fail(c, WS_ERR_MAX_TOO_LOW, "MAX must be > ALWAYS_FILL (lib.rs:1144)");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->counts, 2 * PREPROC_BLOCKS * sizeof(double)))) return rc;
  HIP_TRY(c, preprocess(c->stream, d_data, dtype, n, max_value, (double *)c->counts.p, d_out));
  return WS_OK;
}

int ws_pre_processor(ws_ctx *c, const void *data, int dtype, size_t n, uint8_t max_value, uint8_t *out) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || (n && (!data || !out))) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  const size_t es = preproc_elem_size(dtype);
  if (es == 0) return fail(c, WS_ERR_BAD_ARG, "unknown dtype");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->aux, (n ? n : 1) * es))) return rc;
  if ((rc = ensure(c, c->img, n ? n : 1))) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(c->aux.p, data, n * es, hipMemcpyHostToDevice, c->stream));
  if ((rc = ws_pre_processor_device(c, c->aux.p, dtype, n, max_value, (uint8_t *)c->img.p))) return rc;
  if (n) HIP_TRY(c, hipMemcpyAsync(out, c->img.p, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return WS_OK;
}

}  // extern "C"
