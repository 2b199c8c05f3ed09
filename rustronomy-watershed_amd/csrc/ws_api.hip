// ws_api.hip -- the extern "C" boundary (include/ws_hip.h) and the host-side drivers.
//
// The drivers restate the reference's two transform_with_hook bodies (lib.rs:1328-1522,
// 1638-1808) as launch sequences on one HIP stream.  There is no CPU fallback: every
// entry point that computes needs a HIP device and fails with WS_ERR_NO_DEVICE /
// WS_ERR_HIP otherwise.
#include "../../include/ws_hip.h"
#include "ws_common.hpp"
#include "ws_merge.hpp"
#include "ws_preproc.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <string>
#include <vector>

using namespace wsk;

namespace {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

enum KClass { KC_RELAX = 0, KC_RESOLVE = 1, KC_SWEEP = 2, KC_OTHER = 3, KC_COUNT = 4 };

struct TimedSpan {
  hipEvent_t a, b;
  int cls;
};

}  // namespace

struct ws_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool profiling = false;
  std::string err;
  ws_stats stats{};

  DevBuf img, keys, labels, labels2, stamps, flags, seeds, seeds64, out64, counts, aux, seed_stack;
  DevBuf uf_parent, uf_size, uf_hooked, uf_death, px_items, edge_items, mflags, lakes, refs, seed_tab, tile_list;
  uint32_t *pinned = nullptr;      // FLAG_WORDS words of pinned host memory: the host's mirror of the flag block
  uint32_t *pinned_dev = nullptr;  // the same words as the device sees them (nullptr: not mapped, copies only)
  hipEvent_t ring_ev[COUNTER_RING]{};   // flag slot copied to the host
  hipEvent_t kern_ev[COUNTER_RING]{};   // pass kernel finished
  hipStream_t copy_stream = nullptr;    // carries the per-pass flag read-backs
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t async_ev = nullptr;      // end of the graph a ws_segment_device_begin left in flight
  bool stats_no_wait = false;         // ws_segment_device_end: the stream may hold another context's work behind ours
  bool graph_sufficed = false;        // the last run_fused_form: replayed graph, at its fixpoint after the graph's passes, tables valid
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<TimedSpan> spans;
  std::vector<uint64_t> host64;    // hook staging
  std::vector<uint8_t> host_img;
  size_t last_h = 0, last_w = 0;
  bool have_keys = false;
  bool misc_clean = false;      // the error words of the flag block (FLAG_NERR of them) are known to be zero
  bool expect_sorted = true;    // the last seed list was strictly increasing: try the side-table form first
  uint32_t *tile_min_out = nullptr;   // merging, final labels: run_fused lets the resolve kernel classify the 64x64 tiles into here
  bool tile_min_filled = false;
  bool block_ready = false;       // ws_block_begin has built the seed tables of a row block of block_h x block_w pixels
  size_t block_h = 0, block_w = 0;
  size_t batch_max_px = 0x7FFFFFFFull;      // largest stack of slices run as one transform (ws_ctx_set_batch_pixel_limit)
  size_t seam_min_px = (size_t)1 << 24;     // smallest plane whose pass 1 is a seam repair (ws_ctx_set_seam_repair_min_pixels)
  // ws_segment_device_begin / _end: a transform whose replayed graph has been launched and whose host half (the wait, the
  // look at the convergence and error words, more passes if the flood needs them) is still to come
  enum AsyncPhase { ASYNC_NONE = 0, ASYNC_BEGIN, ASYNC_LAUNCHED, ASYNC_DONE, ASYNC_RESUME };
  int async_phase = ASYNC_NONE;
  int async_rc = 0;
  bool async_merge = false;      // the transform in flight was begun with ws_merge_device_begin
  struct { const uint8_t *d_img; size_t h, w, stride; const uint32_t *d_seeds; size_t n_seeds; ws_options opt; uint32_t *d_labels; } async_args = {};
  uint32_t debug_max_iters = 0xFFFFFFFFu;   // WS_DEBUG_MAXIT: timing experiments only (results wrong when it bites)
  // the optimistic part of a transform (seed tables, first passes, gated resolve, read-backs) as a replayable graph
  struct GraphKey {
    const void *img = nullptr, *seeds = nullptr, *labels = nullptr, *slice_first = nullptr, *tile_min = nullptr;
    size_t stride = 0, n_seeds = 0;
    int ph = 0, pw = 0, slice_h = 0;
    bool padded = false;
    uint32_t max_level = 0;
    uint64_t generation = 0;      // of the context's own buffers (buffer_generation)
    bool operator==(const GraphKey &o) const {
      return generation == o.generation && img == o.img && seeds == o.seeds && labels == o.labels && slice_first == o.slice_first && tile_min == o.tile_min &&
             stride == o.stride && n_seeds == o.n_seeds && ph == o.ph && pw == o.pw && slice_h == o.slice_h && padded == o.padded && max_level == o.max_level;
    }
  };
  GraphKey graph_key, seen_key;      // of graph_exec / of the previous transform
  // the per-level loop of transform_to_list / the merging final labels, captured in groups of levels (merge_host)
  struct ListKey {
    bool merging = false, want_list = false;
    uint32_t levels = 0;
    size_t n_colours = 0, n = 0, cap = 0;
    const void *records = nullptr;      // the buffer the lake records go to (the context's, or a caller's device buffer)
    uint64_t generation = 0;
    bool operator==(const ListKey &o) const {
      return generation == o.generation && generation != 0 && merging == o.merging && want_list == o.want_list && levels == o.levels &&
             n_colours == o.n_colours && n == o.n && cap == o.cap && records == o.records;
    }
  };
  ListKey list_graph_key, list_seen_key;
  hipGraphExec_t list_graphs[16]{};
  hipGraphExec_t graph_exec = nullptr;
  bool graph_unusable = false;       // capture failed once on this stream: not tried again
  uint64_t buffer_generation = 1;    // bumped whenever a device buffer of the context is reallocated
};

static_assert(sizeof(ws_options) == 8, "ws_options is part of the ABI (version 2)");
static_assert(sizeof(ws_stats) == 72, "ws_stats is part of the ABI: graph_launches sits in what was tail padding");

namespace {

// flags buffer layout (u32 words); the pinned host mirror uses the same offsets
constexpr int FLAG_EDGE = 0;                                 // [COUNTER_RING][FLAG_SLOT] striped "a tile edge changed"
constexpr int FLAG_ANY = COUNTER_RING * FLAG_SLOT;           // [FLAG_SLOT] striped "any pixel changed"
constexpr int FLAG_STATS = FLAG_ANY + FLAG_SLOT;             // [2][FLAG_SLOT] striped tile / sweep counters (profiling)
constexpr int FLAG_REFS = FLAG_STATS + 2 * FLAG_SLOT;        // [FLAG_SLOT] striped lengths of the reference work lists
constexpr int FLAG_MISC = FLAG_REFS + FLAG_SLOT;
constexpr int FLAG_OVERFLOW = FLAG_MISC + 0;
constexpr int FLAG_SEED_ERR = FLAG_MISC + 1;
constexpr int FLAG_UNSORTED = FLAG_MISC + 2;                 // seed list not sorted by pixel index
constexpr int FLAG_NONSTRICT = FLAG_MISC + 3;                // seed list not STRICTLY increasing (painting kernels only)
static_assert(FLAG_UNSORTED == FLAG_SEED_ERR + 1 && FLAG_NONSTRICT == FLAG_SEED_ERR + 2, "the seed kernels write these words through one pointer");
constexpr int FLAG_NERR = 4;                                 // OVERFLOW .. NONSTRICT: raised by kernels, never cleared by them
constexpr int FLAG_TOTAL = FLAG_MISC + 4;                    // minima total
constexpr int FLAG_SWEEP = FLAG_MISC + 5;                    // sweep engine: tiles coloured in the last step
constexpr int FLAG_WORDS = FLAG_MISC + 16;

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

int fail(ws_ctx *c, int code, const char *what, hipError_t e = hipSuccess) {
  if (c) {
    c->err = what;
    if (e != hipSuccess) {
      c->err += ": ";
      c->err += hipGetErrorString(e);
    }
  }
  return code;
}

#define HIP_TRY(ctx, call)                                              \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) return fail((ctx), e_ == hipErrorOutOfMemory ? WS_ERR_OOM : WS_ERR_HIP, #call, e_); \
  } while (0)

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

struct Span {
  ws_ctx *c;
  hipEvent_t a = nullptr, b = nullptr;
  int cls;
  Span(ws_ctx *ctx, int k) : c(ctx), cls(k) {
    if (c->profiling) {
      a = next_event(c);
      b = next_event(c);
      if (a) (void)hipEventRecord(a, c->stream);
    }
  }
  ~Span() {
    if (c->profiling && a && b) {
      (void)hipEventRecord(b, c->stream);
      c->spans.push_back({a, b, cls});
    }
  }
};

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

// ---- fused engine ------------------------------------------------------------------------

// Runs `launch(pass)` until a pass reports zero changed tile edges.  Passes are launched in GROUPS:
// one event record + one flag read-back (on a side stream) per group, not per pass -- an event
// record between two dependent kernels costs ~10 us of dependency gap on this stack, a pass that
// has nothing to do ~4 us.  The host stays one group ahead of the flags it reads, so the stream
// never waits for a host round trip.  Groups grow (1, 2, 4, 8, 16): a smooth map needs hundreds of short
// passes, and with groups of two the host's ~100 us of API calls per group was most of their time.
// Flag slots live in a ring of COUNTER_RING passes (pass q clears the slot of pass q + 1): two groups
// in flight must stay below it.
constexpr uint32_t PASS_GROUP_MAX = 16;
// `speculate(last_pass)` (optional) is called once, right after the first lookahead group is queued: work that is only
// valid at the fixpoint may be queued there behind pass `last_pass`, gated on the device by that pass's convergence slot
// (edge_slot()); `*converged_at` = the first pass found to have changed nothing, so the caller can tell whether the gate
// was open (converged_at <= last_pass: a pass after a clean pass is clean).
inline const uint32_t *edge_slot(const uint32_t *d_flags, uint32_t pass);
template <class F>
int pass_loop(ws_ctx *c, uint32_t *d_flags, size_t ntiles, uint32_t *passes_out, F launch, bool zeroed = false,
              uint32_t first_group = 2, const std::function<int(uint32_t)> &speculate = nullptr, uint32_t *converged_at = nullptr,
              uint32_t first_pass = 0) {
  static_assert(2 * PASS_GROUP_MAX < COUNTER_RING, "groups in flight must fit the flag ring");
  if (first_group > PASS_GROUP_MAX) first_group = PASS_GROUP_MAX;
  if (!zeroed) {        // the tile-edge stamps and the convergence ring start at zero
    HIP_TRY(c, hipMemsetAsync(c->stamps.p, 0, ntiles * 4 * 2 * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(d_flags + FLAG_EDGE, 0, COUNTER_RING * FLAG_SLOT * sizeof(uint32_t), c->stream));
  }
  uint32_t launched = first_pass, group = 0;
  struct Group { uint32_t lo, hi, ev; };
  auto launch_group = [&](uint32_t count, Group *g) -> int {
    g->lo = launched;
    for (uint32_t i = 0; i < count; ++i) HIP_TRY(c, launch(launched++));
    g->hi = launched;
    g->ev = group++ % COUNTER_RING;
    // the flag read-back rides a side stream: the next group never queues behind a copy
    HIP_TRY(c, hipEventRecord(c->kern_ev[g->ev], c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->kern_ev[g->ev], 0));
    for (uint32_t p = g->lo; p < g->hi;) {      // the group's slots: one copy, two when they wrap around the ring
      const uint32_t slot = p % COUNTER_RING;
      const uint32_t run = std::min<uint32_t>(g->hi - p, COUNTER_RING - slot);
      HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_EDGE + slot * FLAG_SLOT], d_flags + FLAG_EDGE + slot * FLAG_SLOT,
                                (size_t)run * FLAG_SLOT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->copy_stream));
      p += run;
    }
    HIP_TRY(c, hipEventRecord(c->ring_ev[g->ev], c->copy_stream));
    return WS_OK;
  };
  Group done{}, ahead{};
  int rc;
  if ((rc = launch_group(first_group, &done))) return rc;
  uint32_t size = 1;          // the first lookahead group: one pass is enough to keep the stream busy while the host reads
  bool first = true;
  for (;;) {
    if ((rc = launch_group(size, &ahead))) return rc;
    if (first && speculate && (rc = speculate(launched - 1))) return rc;
    first = false;
    HIP_TRY(c, hipEventSynchronize(c->ring_ev[done.ev]));
    bool converged = false;
    for (uint32_t p = done.lo; p < done.hi && !converged; ++p) {
      converged = !slot_nonzero(&c->pinned[FLAG_EDGE + (p % COUNTER_RING) * FLAG_SLOT]);
      if (converged && converged_at) *converged_at = p;
    }
    if (converged) break;
    done = ahead;
    size = std::min(size * 2, PASS_GROUP_MAX);
  }
  // later work on the main stream may reuse the flag words: order it after the last read-back
  HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ring_ev[ahead.ev], 0));
  *passes_out = launched;
  return WS_OK;
}

constexpr int WS_INTERNAL_PENDING = 0x7fff;      // run_fused_form -> ws_segment_device_begin: graph launched, host half pending (never leaves the library)
constexpr uint32_t GRAPH_PASSES = 5;      // passes inside the graph; the resolve is gated on the last one (the bench field: pass 3 still moves a few tiles, pass 4 finds nothing -- a sixth pass was 6 us of idle launch)

inline const uint32_t *edge_slot(const uint32_t *d_flags, uint32_t pass) {
  return d_flags + FLAG_EDGE + (size_t)(pass % COUNTER_RING) * FLAG_SLOT;
}

// Seeds reach the kernels in one of two forms.  TABLES: a strictly increasing list (what
// find_local_minima returns) is turned into one bit per pixel plus a list index per 32-pixel word
// (seed_tables), 16 MiB at 8192^2; relaxation pass 0 reads the bits, the resolve kernel computes seed
// colours from both and writes the label plane exactly once.  PAINTED: any list -- the label plane is
// painted first (paint_labels) and read back twice.  Whether a list is strictly increasing is only
// known on the device, so the choice is a prediction: the context tries TABLES while the previous
// list was strictly increasing; a wrong guess is detected by the table builder itself
// (FLAG_NONSTRICT), costs one wasted transform, and flips the prediction.
// slice_h > 0: the plane is a stack of ph / slice_h independent slices (ws_segment_batch_device); d_seeds are then in
// stacked coordinates and slice_first (device) holds every slice's first list index, so that colours restart per slice.
int run_fused_form(ws_ctx *c, const uint8_t *d_img, size_t stride, int ph, int pw, uint32_t max_level,
                   const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, bool tables, bool *mispredicted,
                   int slice_h = 0, const uint32_t *slice_first = nullptr, bool padded = false) {
  // padded: edge correction -- d_img is the caller's (ph - 2) x (pw - 2) image (per slice), the ring of zeros around it is
  // virtual (padded_img_index, ws_common.hpp)
  const size_t n = (size_t)ph * pw;
  const size_t ntiles = (size_t)tiles_of(pw) * tiles_of(ph);
  const size_t nwords = (n + 31) / 32;
  int rc;
  if ((rc = ensure(c, c->keys, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->stamps, std::max(ntiles, relax_tiles(ph, pw)) * 4 * 2 * sizeof(uint32_t)))) return rc;
  if (tables && (rc = ensure(c, c->seed_tab, (nwords ? nwords : 1) * 2 * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->tile_list, relax_list_words(ph, pw) * sizeof(uint32_t)))) return rc;
  uint32_t *tile_list = (uint32_t *)c->tile_list.p;
  uint32_t *keys = (uint32_t *)c->keys.p;
  uint32_t *flags = (uint32_t *)c->flags.p;
  uint32_t *stamps = (uint32_t *)c->stamps.p;
  uint32_t *seed_mask = tables ? (uint32_t *)c->seed_tab.p : nullptr, *word_base = tables ? seed_mask + nwords : nullptr;
  c->have_keys = false;
  *mispredicted = false;
  c->graph_sufficed = false;

  // ---- graph replay -------------------------------------------------------------------------------
  // A transform that repeats the previous one's arguments exactly (same buffers, sizes and seed COUNT; the contents
  // are free to change: a pipeline that reuses its buffers) replays its optimistic part -- seed tables, the first
  // GRAPH_PASSES passes, the gated resolve, the read-backs -- as one hipGraph launch instead of eleven stream
  // operations: the second such transform captures it, later ones replay (1024^2: 0.141 -> 0.100 ms, 2048^2: 0.162 ->
  // 0.133 ms, 8192^2: -2 %).  The host then looks at the lookahead pass's slot; a flood that needs more passes goes on
  // with the ordinary loop.  Not on the legacy null stream (capture is not allowed there).
  static const bool use_graph = tuning_env("WS_NO_GRAPH") == nullptr;      // A/B knob for tools/
  int graph_mode = 0;      // 1: replayed, 2: captured now
  ws_ctx::GraphKey key;
  key.img = d_img; key.seeds = d_seeds; key.labels = d_labels; key.slice_first = slice_first; key.tile_min = c->tile_min_out;
  key.stride = stride; key.n_seeds = n_seeds; key.ph = ph; key.pw = pw; key.slice_h = slice_h; key.padded = padded; key.max_level = max_level;
  key.generation = c->buffer_generation;
  const bool graph_ok = use_graph && c->stream != nullptr && !c->graph_unusable && tables && n != 0 && n < 0x80000000ull && !c->profiling && c->misc_clean &&
                        c->debug_max_iters == 0xFFFFFFFFu;
  const bool resume = c->async_phase == ws_ctx::ASYNC_RESUME;      // ws_segment_device_end: the graph of this very call is in flight
  if (resume) c->async_phase = ws_ctx::ASYNC_NONE;      // (consumed: a repeat with painted seeds after a wrong guess is an ordinary run)
  if (resume) graph_mode = 1;
  else if (graph_ok && c->graph_exec && key == c->graph_key) graph_mode = 1;
  else if (graph_ok && key == c->seen_key) graph_mode = 2;
  if (!resume) c->seen_key = graph_ok ? key : ws_ctx::GraphKey();
  if (graph_mode != 0) {
    if ((rc = ensure(c, c->refs, resolve_ref_capacity(ph, pw) * sizeof(uint32_t)))) return rc;
    if (c->buffer_generation != key.generation && !resume) graph_mode = 0;      // that allocation moved a buffer: next time
  }
  if (graph_mode == 2) {
    if (c->graph_exec) { (void)hipGraphExecDestroy(c->graph_exec); c->graph_exec = nullptr; }
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
      (void)hipGetLastError();
      c->graph_unusable = true;
      graph_mode = 0;
    }
  }
  if (graph_mode == 2) {
    const PassFlags gpf = make_pf(c);
    hipGraph_t graph = nullptr;
    hipError_t e = seed_tables(c->stream, d_seeds, n_seeds, ph, pw, seed_mask, word_base, flags + FLAG_SEED_ERR, stamps,
                               relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC, slice_first, (size_t)slice_h * pw);
    for (uint32_t pass = 0; pass < GRAPH_PASSES && e == hipSuccess; ++pass)
      e = relax_pass(c->stream, d_img, stride, keys, ph, pw, max_level, pass, stamps, gpf, c->debug_max_iters, seed_mask, true, slice_h, true, padded, tile_list, c->seam_min_px);
    const uint32_t last = GRAPH_PASSES - 1;
    if (e == hipSuccess)
      e = resolve_two_launch(c->stream, keys, d_labels, ph, pw, (uint32_t *)c->refs.p, c->debug_max_iters, seed_mask, word_base,
                             c->tile_min_out, edge_slot(flags, last), slice_h, flags + FLAG_OVERFLOW, flags + FLAG_SEED_ERR);
    // the read-backs: the lookahead pass's convergence slot and the error words
    if (e == hipSuccess && c->pinned_dev)
      e = words_to_host(c->stream, edge_slot(flags, last), FLAG_SLOT, c->pinned_dev + FLAG_EDGE + (last % COUNTER_RING) * FLAG_SLOT,
                        flags + FLAG_OVERFLOW, FLAG_NERR, c->pinned_dev + FLAG_OVERFLOW);
    if (e == hipSuccess && !c->pinned_dev)
      e = hipMemcpyAsync(&c->pinned[FLAG_EDGE + (last % COUNTER_RING) * FLAG_SLOT], edge_slot(flags, last), FLAG_SLOT * sizeof(uint32_t),
                         hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && !c->pinned_dev)
      e = hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, FLAG_NERR * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
    const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
    if (e == hipSuccess && e2 == hipSuccess) e = hipGraphInstantiate(&c->graph_exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess || e2 != hipSuccess) {      // nothing ran: take the ordinary path, for good
      (void)hipGetLastError();
      c->graph_exec = nullptr;
      c->graph_unusable = true;
      graph_mode = 0;
    } else {
      c->graph_key = key;
    }
  }
  if (graph_mode != 0) {
    c->have_keys = false;
    *mispredicted = false;
    c->misc_clean = false;
    c->stats.graph_launches = 1;
    if (!resume) HIP_TRY(c, hipGraphLaunch(c->graph_exec, c->stream));
    if (c->async_phase == ws_ctx::ASYNC_BEGIN && graph_mode == 1) {      // ws_segment_device_begin: the host half waits for _end
      HIP_TRY(c, hipEventRecord(c->async_ev, c->stream));
      c->async_phase = ws_ctx::ASYNC_LAUNCHED;
      return WS_INTERNAL_PENDING;
    }
    // (_end waits for the graph's own end, not for the stream: another context may have queued its transform behind it)
    if (resume) HIP_TRY(c, hipEventSynchronize(c->async_ev));
    else HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else {
    Span sp(c, KC_OTHER);
    // The error words (ring overflow, seed out of bounds, list unsorted / not strict) are only ever
    // RAISED by kernels; they are known to be zero after a transform that read them back as zero, and
    // cleared here otherwise -- the seed kernel cannot clear words it may have to raise.
    if (!c->misc_clean) HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, FLAG_NERR * sizeof(uint32_t), c->stream));
    c->misc_clean = false;
    // The same launch clears the relaxation's tile-edge stamps and the striped flag words.  The
    // arrival-stamp plane is not touched: relaxation pass 0 derives it from the seeds.
    if (tables)
      HIP_TRY(c, seed_tables(c->stream, d_seeds, n_seeds, ph, pw, seed_mask, word_base, flags + FLAG_SEED_ERR, stamps,
                             relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC, slice_first, (size_t)slice_h * pw));
    else      // one pass over the label plane paints the seeds (colour i + 1, later duplicates win), zero elsewhere
      HIP_TRY(c, paint_labels(c->stream, d_seeds, n_seeds, ph, pw, d_labels, flags + FLAG_SEED_ERR, stamps,
                              relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC));
  }
  if (n == 0) return WS_OK;

  const PassFlags pf = make_pf(c);
  // The label resolve is queued speculatively behind the first lookahead pass, gated on the device by that pass's
  // convergence slot: when the host then reads that the flood was already at its fixpoint (the bench field: always),
  // the labels are being written while it reads, instead of the GPU idling through the round trip.
  const bool two_launch = n < 0x80000000ull;
  static const bool no_speculation = tuning_env("WS_NO_SPECULATION") != nullptr;      // A/B knob for tools/
  if (two_launch && (rc = ensure(c, c->refs, resolve_ref_capacity(ph, pw) * sizeof(uint32_t)))) return rc;
  auto resolve = [&](const uint32_t *gate) -> int {
    Span sp(c, KC_RESOLVE);
    HIP_TRY(c, resolve_two_launch(c->stream, keys, d_labels, ph, pw, (uint32_t *)c->refs.p, c->debug_max_iters, seed_mask, word_base,
                                  c->tile_min_out, gate, slice_h, flags + FLAG_OVERFLOW, flags + FLAG_SEED_ERR));
    return WS_OK;
  };
  uint32_t speculated_after = 0xFFFFFFFFu, converged_at = 0xFFFFFFFFu;
  std::function<int(uint32_t)> speculate = nullptr;
  if (two_launch && !no_speculation)
    speculate = [&](uint32_t last_pass) -> int {
      speculated_after = last_pass;
      return resolve(edge_slot(flags, last_pass));
    };
  auto launch_pass = [&](uint32_t pass) {
    Span sp(c, KC_RELAX);
    return relax_pass(c->stream, d_img, stride, keys, ph, pw, max_level, pass, stamps, pf, c->debug_max_iters,
                      tables ? seed_mask : d_labels, tables, slice_h, two_launch, padded, tile_list, c->seam_min_px);
  };
  if (graph_mode != 0) {
    // the graph ran seed tables, passes 0 .. GRAPH_PASSES - 1, the gated resolve and the read-backs
    speculated_after = GRAPH_PASSES - 1;
    if (slot_nonzero(&c->pinned[FLAG_EDGE + ((GRAPH_PASSES - 1) % COUNTER_RING) * FLAG_SLOT])) {
      rc = pass_loop(c, flags, relax_tiles(ph, pw), &c->stats.relax_passes, launch_pass, true, 2, nullptr, &converged_at, GRAPH_PASSES);
      if (rc) return rc;
    } else {
      converged_at = GRAPH_PASSES - 1;
      c->stats.relax_passes = GRAPH_PASSES;
    }
  } else {
    rc = pass_loop(c, flags, relax_tiles(ph, pw), &c->stats.relax_passes, launch_pass, true, 5, speculate, &converged_at);
    if (rc) return rc;
  }
  // (a seam repair is two launches for pass 1: bands, strips)
  c->stats.launches_relax = c->stats.relax_passes + (c->stats.relax_passes >= 2 && relax_uses_seam_repair(ph, pw, tables, slice_h, padded, c->seam_min_px) ? 1u : 0u);

  // no host round trip here: the error words are read once, after the resolve launches are queued
  if (two_launch) {
    const bool already = speculated_after != 0xFFFFFFFFu && converged_at <= speculated_after;     // the gate was open
    if (!already && (rc = resolve(nullptr))) return rc;
    c->tile_min_filled = c->tile_min_out != nullptr;
    c->stats.resolve_passes = 2;
  } else {
    rc = pass_loop(c, flags, ntiles, &c->stats.resolve_passes, [&](uint32_t pass) {
      Span sp(c, KC_RESOLVE);
      return resolve_pass(c->stream, keys, d_labels, ph, pw, pass, stamps, pf);
    });
    if (rc) return rc;
  }
  c->stats.launches_resolve = c->stats.resolve_passes;
  const bool all_read = graph_mode != 0 && converged_at == GRAPH_PASSES - 1;      // the graph's own read-backs cover everything
  if (!all_read)
    HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, FLAG_NERR * sizeof(uint32_t),
                              hipMemcpyDeviceToHost, c->stream));
  if (all_read) {
  } else if (c->profiling) {      // striped statistics: tiles that ran and in-tile sweeps, summed over passes
    HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_STATS], flags + FLAG_STATS, 2 * FLAG_SLOT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t quarter_tiles = 0;      // k_relax counts 2048-pixel units: the seam repair's bands are quarter tiles
    for (int i = 0; i < NSTRIPE; ++i) {
      quarter_tiles += c->pinned[FLAG_STATS + i * STRIPE_STRIDE];
      c->stats.relax_tile_iterations += c->pinned[FLAG_STATS + FLAG_SLOT + i * STRIPE_STRIDE];
    }
    c->stats.tiles_run_relax += (quarter_tiles + 2) / 4;
  } else {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (c->pinned[FLAG_SEED_ERR]) return fail(c, WS_ERR_SEED_OOB, "seed outside the label plane (the reference panics: lib.rs:1676)");
  c->expect_sorted = c->pinned[FLAG_NONSTRICT] == 0;
  if (tables && !c->expect_sorted) {      // the tables describe some other list: nothing computed from them counts
    *mispredicted = true;
    return WS_OK;
  }
  if (c->pinned[FLAG_OVERFLOW]) return fail(c, WS_ERR_RING_OVERFLOW, "more than 2^24-1 flood rings inside one level");
  c->misc_clean = c->pinned[FLAG_UNSORTED] == 0 && c->pinned[FLAG_NONSTRICT] == 0;
  c->graph_sufficed = graph_mode != 0 && converged_at == GRAPH_PASSES - 1;
  c->have_keys = true;
  c->last_h = ph;
  c->last_w = pw;
  return WS_OK;
}

int run_fused(ws_ctx *c, const uint8_t *d_img, size_t stride, int ph, int pw, uint32_t max_level,
              const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, bool padded = false) {
  // the side-table form needs nibble-aligned patch rows (W % 4 == 0) and the two-launch resolve
  static const bool no_tables = tuning_env("WS_NO_SEED_TABLES") != nullptr;      // A/B knob for tools/
  const bool can_tables = !no_tables && (pw & 3) == 0 && (size_t)ph * pw < 0x80000000ull && n_seeds > 0;
  bool mispredicted = false;
  int rc = run_fused_form(c, d_img, stride, ph, pw, max_level, d_seeds, n_seeds, d_labels, can_tables && c->expect_sorted, &mispredicted, 0, nullptr, padded);
  if (rc == WS_OK && mispredicted) rc = run_fused_form(c, d_img, stride, ph, pw, max_level, d_seeds, n_seeds, d_labels, false, &mispredicted, 0, nullptr, padded);
  return rc;
}

// ---- sweep engine ------------------------------------------------------------------------

// lib.rs:1689-1748 literally: for every level, flood steps until one colours nothing.
// `after_level` (optional) sees the plane after each level's loop (the hook point).
template <class F>
int run_sweep(ws_ctx *c, const uint8_t *d_img, size_t stride, int ph, int pw, uint32_t max_level,
              const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, F after_level, bool padded = false) {
  const size_t n = (size_t)ph * pw;
  int rc;
  if ((rc = ensure(c, c->labels2, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  uint32_t *flags = (uint32_t *)c->flags.p;
  uint32_t *cur = d_labels, *nxt = (uint32_t *)c->labels2.p;
  c->have_keys = false;
  HIP_TRY(c, hipMemsetAsync(cur, 0, n * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, 3 * sizeof(uint32_t), c->stream));
  c->misc_clean = false;
  HIP_TRY(c, scatter_seeds(c->stream, d_seeds, nullptr, n_seeds, ph, pw, cur, nullptr, flags + FLAG_SEED_ERR));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, 2 * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_SEED_ERR]) return fail(c, WS_ERR_SEED_OOB, "seed outside the label plane (the reference panics: lib.rs:1676)");

  for (uint32_t lvl = 0; lvl <= max_level; ++lvl) {
    for (;;) {
      if (n == 0) break;
      {
        Span sp(c, KC_SWEEP);
        HIP_TRY(c, hipMemsetAsync(flags + FLAG_SWEEP, 0, sizeof(uint32_t), c->stream));
        HIP_TRY(c, flood_step(c->stream, d_img, stride, cur, nxt, ph, pw, lvl, flags + FLAG_SWEEP, padded));
        c->stats.sweep_steps++;
      }
      HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_SWEEP], flags + FLAG_SWEEP, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      std::swap(cur, nxt);                 // an empty step copies the plane, so either buffer is current
      if (c->pinned[FLAG_SWEEP] == 0) break;   // lib.rs:1733-1735
    }
    rc = after_level(lvl, cur);
    if (rc) return rc;
  }
  c->stats.launches_sweep = c->stats.sweep_steps;
  if (cur != d_labels && n)
    HIP_TRY(c, hipMemcpyAsync(d_labels, cur, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  return WS_OK;
}

int pick_engine(const ws_options *opt) {
  return opt->engine == WS_ENGINE_SWEEP ? WS_ENGINE_SWEEP : WS_ENGINE_FUSED;
}

// With edge correction the kernels read the caller's own image through a virtual ring of zeros (padded_img_index):
// no padded copy exists.  An EMPTY image still has a plane (2 x (w + 2) border pixels, never flooded) and the kernels
// read a clamped address for border pixels, which must exist: a zeroed block of the context stands in.
int empty_image_block(ws_ctx *c, const uint8_t **d_img, size_t *d_stride) {
  int rc;
  if ((rc = ensure(c, c->img, 16))) return rc;
  HIP_TRY(c, hipMemsetAsync(c->img.p, 0, 16, c->stream));
  *d_img = (const uint8_t *)c->img.p;
  *d_stride = 1;
  return WS_OK;
}

// lib.rs:1675-1677 indexes the padded plane with the caller's coordinates (seed_shift 0); seed_shift 1 moves every seed
// by (+1, +1), onto the pixel it was found at.  Only meaningful with edge correction.
inline uint32_t seed_shift_of(const ws_options *opt) { return opt->edge_correction && opt->seed_shift ? 1u : 0u; }

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

  if (h * w == 0) return empty_image_block(c, d_img, d_stride);
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

int segment_host(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                 size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels) {
  if (!c) return WS_ERR_BAD_ARG;
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = ph * pw;
  const uint8_t *d_img;
  size_t d_stride;
  const uint32_t *d_seeds;
  if ((rc = ensure(c, c->labels, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->out64, (n ? n : 1) * sizeof(uint64_t)))) return rc;
  stats_begin(c);
  if ((rc = stage_inputs(c, img, h, w, stride, seeds_rc, n_seeds, opt, ph, pw, &d_img, &d_stride, &d_seeds))) return rc;
  uint32_t *d_labels = (uint32_t *)c->labels.p;
  uint64_t *d_out64 = (uint64_t *)c->out64.p;
  const uint8_t *himg = cb ? hook_image(c, img, h, w, stride, opt->edge_correction) : nullptr;
  if (cb) c->host64.resize(n ? n : 1);

  if (pick_engine(opt) == WS_ENGINE_SWEEP) {
    rc = run_sweep(c, d_img, d_stride, (int)ph, (int)pw, opt->max_water_level, d_seeds, n_seeds, d_labels,
                   [&](uint32_t lvl, const uint32_t *cur) -> int {
                     if (!cb) return WS_OK;
                     HIP_TRY(c, widen_labels(c->stream, cur, d_out64, n));
                     HIP_TRY(c, hipMemcpyAsync(c->host64.data(), d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
                     HIP_TRY(c, hipStreamSynchronize(c->stream));
                     cb(user, (uint8_t)lvl, opt->max_water_level, himg, c->host64.data(), ph, pw);   // lib.rs:1796-1804
                     return WS_OK;
                   }, opt->edge_correction != 0);
    if (rc) return rc;
  } else {
    rc = run_fused(c, d_img, d_stride, (int)ph, (int)pw, opt->max_water_level, d_seeds, n_seeds, d_labels, opt->edge_correction != 0);
    if (rc) return rc;
    if (cb) {
      for (uint32_t lvl = 0; lvl <= opt->max_water_level; ++lvl) {
        HIP_TRY(c, snapshot_level(c->stream, (const uint32_t *)c->keys.p, d_labels, d_out64, n, lvl));
        HIP_TRY(c, hipMemcpyAsync(c->host64.data(), d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        cb(user, (uint8_t)lvl, opt->max_water_level, himg, c->host64.data(), ph, pw);
      }
    }
  }
  if (out_labels && n) {
    Span sp(c, KC_OTHER);
    HIP_TRY(c, widen_labels(c->stream, d_labels, d_out64, n));
    HIP_TRY(c, hipMemcpyAsync(out_labels, d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  }
  return stats_end(c);
}

}  // namespace

// ============================================================================ C ABI ====

extern "C" {

// a context that holds a transform begun with ws_*_device_begin takes no other work until the matching _end
static int refuse_if_in_flight(ws_ctx *c) {
  if (c && c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "the context holds a transform begun with ws_*_device_begin: end it first");
  return WS_OK;
}

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
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (DevBuf *b : {&c->img, &c->keys, &c->labels, &c->labels2, &c->stamps, &c->flags, &c->seeds, &c->out64, &c->counts, &c->aux, &c->seed_stack, &c->seeds64,
                    &c->uf_parent, &c->uf_size, &c->uf_hooked, &c->uf_death, &c->px_items, &c->edge_items, &c->mflags, &c->lakes, &c->refs, &c->seed_tab, &c->tile_list})
    if (b->p) (void)hipFree(b->p);
  if (c->pinned) (void)hipHostFree(c->pinned);
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
  if (!c) return WS_ERR_BAD_ARG;
  c->batch_max_px = max_px == 0 ? 0x7FFFFFFFull : std::min<size_t>(max_px, 0x7FFFFFFFull);
  return WS_OK;
}

int ws_ctx_set_seam_repair_min_pixels(ws_ctx *c, size_t min_px) {
  if (!c) return WS_ERR_BAD_ARG;
  c->seam_min_px = min_px == 0 ? (size_t)1 << 24 : min_px;
  ++c->buffer_generation;      // a captured graph holds the launches of the other flow
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
  HIP_TRY(c, exclusive_scan_u32(c->stream, counts, nseg, flags + FLAG_TOTAL));
  HIP_TRY(c, minima_write(c->stream, nibbles, (int)h, (int)w, counts, d_out_rc, cap));
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
  if ((rc = ensure(c, c->out64, (dcap ? dcap : 1) * 2 * sizeof(uint64_t)))) return rc;
  HIP_TRY(c, hipMemcpy2DAsync(c->img.p, w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  rc = ws_find_local_minima_device(c, (const uint8_t *)c->img.p, h, w, w, (uint32_t *)c->seeds.p, dcap, n_found);
  if (rc != WS_OK && rc != WS_ERR_CAPACITY) return rc;
  const size_t got = std::min(*n_found, dcap);
  if (got) {
    HIP_TRY(c, widen_pairs(c->stream, (const uint32_t *)c->seeds.p, (uint64_t *)c->out64.p, got * 2));
    HIP_TRY(c, hipMemcpyAsync(out_rc, c->out64.p, got * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  return rc;
}

// ---- segmenting ---------------------------------------------------------------------------

int ws_segment(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
               size_t n_seeds, const ws_options *opt, uint64_t *out_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!out_labels) return fail(c, WS_ERR_BAD_ARG, "out_labels is null");
  return segment_host(c, img, h, w, stride, seeds_rc, n_seeds, opt, nullptr, nullptr, out_labels);
}

int ws_segment_with_hook(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                         size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  return segment_host(c, img, h, w, stride, seeds_rc, n_seeds, opt, cb, user, out_labels);
}

static int segment_device_body(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                               size_t n_seeds, const ws_options *opt, uint32_t *d_labels) {
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  if ((!d_img && h * w) || (!d_seeds_rc && n_seeds) || (!d_labels && ph * pw)) return fail(c, WS_ERR_BAD_ARG, "null device pointer");
  if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
  HIP_TRY(c, hipSetDevice(c->device));
  stats_begin(c);
  const uint8_t *src = d_img;
  size_t src_stride = stride;
  const bool padded = opt->edge_correction != 0;      // the ring of zeros is virtual: no padded copy
  if (padded && h * w == 0 && (rc = empty_image_block(c, &src, &src_stride))) return rc;
  const uint32_t *seeds;
  if ((rc = shifted_seeds(c, d_seeds_rc, n_seeds, opt, &seeds))) return rc;
  if (pick_engine(opt) == WS_ENGINE_SWEEP)
    rc = run_sweep(c, src, src_stride, (int)ph, (int)pw, opt->max_water_level, seeds, n_seeds, d_labels,
                   [](uint32_t, const uint32_t *) { return (int)WS_OK; }, padded);
  else
    rc = run_fused(c, src, src_stride, (int)ph, (int)pw, opt->max_water_level, seeds, n_seeds, d_labels, padded);
  if (rc) return rc;      // (WS_INTERNAL_PENDING included: ws_segment_device_begin)
  return stats_end(c);
}

int ws_segment_device(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                      size_t n_seeds, const ws_options *opt, uint32_t *d_labels) {
  if (!c) return WS_ERR_BAD_ARG;
  if (c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "a transform begun with ws_segment_device_begin has not been ended");
  return segment_device_body(c, d_img, h, w, stride, d_seeds_rc, n_seeds, opt, d_labels);
}

// The two halves of ws_segment_device.  _begin queues the transform and returns; _end waits for it and reports its
// status.  What can be queued without the host looking is the replayed graph of a transform that repeats the previous
// one's arguments (run_fused_form): any other call runs whole inside _begin.  Between the two the context belongs to the
// transform: no other call on it, and the caller's buffers must stay as they are (a flood that needs more passes than the
// graph holds goes on inside _end).  Two contexts that take turns keep the GPU's queue from running dry between
// transforms -- the host's wait-and-relaunch is ~15 us of a 0.55 ms transform at 8192^2.
int ws_segment_device_begin(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                            size_t n_seeds, const ws_options *opt, uint32_t *d_labels) {
  if (!c || !opt) return WS_ERR_BAD_ARG;
  if (c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "ws_segment_device_begin: the previous transform has not been ended");
  c->async_args = {d_img, h, w, stride, d_seeds_rc, n_seeds, *opt, d_labels};
  c->async_phase = ws_ctx::ASYNC_BEGIN;
  const int rc = segment_device_body(c, d_img, h, w, stride, d_seeds_rc, n_seeds, opt, d_labels);
  if (rc == WS_INTERNAL_PENDING && c->async_phase == ws_ctx::ASYNC_LAUNCHED) return WS_OK;
  c->async_phase = ws_ctx::ASYNC_DONE;      // ran whole (or failed): _end hands the status over
  c->async_rc = rc == WS_INTERNAL_PENDING ? (int)WS_ERR_UNSUPPORTED : rc;
  return WS_OK;
}

int ws_segment_device_end(ws_ctx *c) {
  if (!c) return WS_ERR_BAD_ARG;
  if (c->async_merge) return fail(c, WS_ERR_BAD_ARG, "ws_segment_device_end: the transform in flight was begun with ws_merge_device_begin");
  if (c->async_phase == ws_ctx::ASYNC_DONE) { c->async_phase = ws_ctx::ASYNC_NONE; return c->async_rc; }
  if (c->async_phase != ws_ctx::ASYNC_LAUNCHED) return fail(c, WS_ERR_BAD_ARG, "ws_segment_device_end without ws_segment_device_begin");
  c->async_phase = ws_ctx::ASYNC_RESUME;
  c->stats_no_wait = true;
  const auto a = c->async_args;
  const int rc = segment_device_body(c, a.d_img, a.h, a.w, a.stride, a.d_seeds, a.n_seeds, &a.opt, a.d_labels);
  c->stats_no_wait = false;
  c->async_phase = ws_ctx::ASYNC_NONE;
  return rc == WS_INTERNAL_PENDING ? (int)WS_ERR_UNSUPPORTED : rc;
}

int ws_last_arrival_device(ws_ctx *c, const uint32_t **d_keys, size_t *h, size_t *w) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_keys || !h || !w) return WS_ERR_BAD_ARG;
  if (!c->have_keys) return fail(c, WS_ERR_UNSUPPORTED, "no arrival stamps: the last call did not use the fused engine");
  *d_keys = (const uint32_t *)c->keys.p;
  *h = c->last_h;
  *w = c->last_w;
  return WS_OK;
}

int ws_copy_last_arrival_device(ws_ctx *c, uint32_t *d_dst, size_t n_elems) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_dst) return WS_ERR_BAD_ARG;
  if (!c->have_keys) return fail(c, WS_ERR_UNSUPPORTED, "no arrival stamps: the last call did not use the fused engine");
  const size_t n = c->last_h * c->last_w;
  if (n_elems < n) return fail(c, WS_ERR_CAPACITY, "arrival buffer too small");
  HIP_TRY(c, hipSetDevice(c->device));
  if (n) HIP_TRY(c, hipMemcpyAsync(d_dst, c->keys.p, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  return WS_OK;
}

int ws_level_snapshot_device(ws_ctx *c, const uint32_t *d_labels, uint8_t water_level, uint32_t *d_out) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_labels || !d_out) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (!c->have_keys) return fail(c, WS_ERR_UNSUPPORTED, "no arrival stamps: the last call was not a fused-engine ws_segment_device / ws_merge_device");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, snapshot_level_u32(c->stream, (const uint32_t *)c->keys.p, d_labels, d_out, c->last_h * c->last_w, water_level));
  return WS_OK;
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

// ---- row blocks of one larger field (multi-GPU tiling) -----------------------------------------
//
// A rank owns a block of rows of the global field and holds it with one extra row on every side
// that has a neighbour rank.  First/last local rows are therefore either the global border or a
// halo copy, i.e. exactly the rows the flood never writes (lib.rs:220-222), so the single-GPU
// kernels run unchanged on the local plane; the caller exchanges halo rows between calls.

int ws_block_init(ws_ctx *c, size_t h, size_t w, const uint32_t *d_seeds_rc, const uint32_t *d_colours, size_t n_seeds,
                  uint32_t *d_keys, uint32_t *d_labels) {
  if (!c || (h * w && (!d_keys || !d_labels)) || (n_seeds && (!d_seeds_rc || !d_colours))) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^32 pixels");
  HIP_TRY(c, hipSetDevice(c->device));
  uint32_t *flags = (uint32_t *)c->flags.p;
  const size_t n = h * w;
  HIP_TRY(c, fill_u32(c->stream, d_keys, n, KEY_INF));
  if (n) HIP_TRY(c, hipMemsetAsync(d_labels, 0, n * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, 3 * sizeof(uint32_t), c->stream));
  c->misc_clean = false;
  HIP_TRY(c, scatter_seeds(c->stream, d_seeds_rc, d_colours, n_seeds, (int)h, (int)w, d_labels, d_keys, flags + FLAG_SEED_ERR));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_SEED_ERR]) return fail(c, WS_ERR_SEED_OOB, "seed outside the local plane");
  return WS_OK;
}

int ws_block_relax(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, uint8_t max_water_level,
                   uint32_t *d_keys, int *changed) {
  if (!c || !changed || (h * w && (!d_img || !d_keys)) || stride < w) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^32 pixels");
  *changed = 0;
  if (h * w == 0) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->stamps, relax_tiles((int)h, (int)w) * 4 * 2 * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->tile_list, relax_list_words((int)h, (int)w) * sizeof(uint32_t)))) return rc;
  uint32_t *tile_list = (uint32_t *)c->tile_list.p;
  uint32_t *flags = (uint32_t *)c->flags.p, *stamps = (uint32_t *)c->stamps.p;
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_ANY, 0, FLAG_SLOT * sizeof(uint32_t), c->stream));
  PassFlags pf = make_pf(c);
  pf.stats = nullptr;
  uint32_t passes = 0;
  rc = pass_loop(c, flags, relax_tiles((int)h, (int)w), &passes, [&](uint32_t pass) {
    return relax_pass(c->stream, d_img, stride, d_keys, (int)h, (int)w, max_water_level, pass, stamps, pf, c->debug_max_iters, nullptr, false, 0, false, false, tile_list);
  });
  if (rc) return rc;
  c->stats.relax_passes += passes;
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_ANY], flags + FLAG_ANY, FLAG_SLOT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_OVERFLOW]) return fail(c, WS_ERR_RING_OVERFLOW, "more than 2^24-1 flood rings inside one level");
  *changed = slot_nonzero(&c->pinned[FLAG_ANY]);
  return WS_OK;
}

int ws_block_resolve(ws_ctx *c, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w, int *changed) {
  if (!c || !changed || (h * w && (!d_keys || !d_labels))) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^32 pixels");
  *changed = 0;
  if (h * w == 0) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t ntiles = (size_t)tiles_of((int)w) * tiles_of((int)h);
  int rc;
  if ((rc = ensure(c, c->stamps, ntiles * 4 * 2 * sizeof(uint32_t)))) return rc;
  uint32_t *flags = (uint32_t *)c->flags.p, *stamps = (uint32_t *)c->stamps.p;
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_ANY, 0, FLAG_SLOT * sizeof(uint32_t), c->stream));
  PassFlags pf = make_pf(c);
  pf.stats = nullptr;
  uint32_t passes = 0;
  rc = pass_loop(c, flags, ntiles, &passes, [&](uint32_t pass) {
    return resolve_pass(c->stream, d_keys, d_labels, (int)h, (int)w, pass, stamps, pf);
  });
  if (rc) return rc;
  c->stats.resolve_passes += passes;
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_ANY], flags + FLAG_ANY, FLAG_SLOT * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *changed = slot_nonzero(&c->pinned[FLAG_ANY]);
  return WS_OK;
}

// ---- row blocks, fast form: seed side tables, speculative passes, two-launch resolve, one table exchange -------------
//
// The same block as above, for seed lists in strictly increasing order (what find_local_minima returns; a rank's seeds
// are then one contiguous range of the caller's list, colours first_colour, first_colour + 1, ...):
//   ws_block_begin            seed tables + relaxation to LOCAL convergence (halo rows hold whatever the caller put there;
//                             before the first exchange: nothing, the seed bits decide)
//   ws_block_relax_halo       after the caller rewrote the halo rows of d_keys: only the tile rows that hold them start,
//                             changes spread from there; again to local convergence
//   ws_block_resolve_local    labels of the whole block in two launches; a chain that ends on a halo pixel stays a
//                             reference to it
//   ws_block_export_boundary  the block's two boundary rows as entries of the global boundary table (ws_block.hip)
//   ws_block_import_boundary  resolves the all-gathered table, writes the halo rows, finishes the chains
// rustronomy-watershed_amd/distributed.py drives the exchange (RCCL through torch.distributed).

static int block_check(ws_ctx *c, size_t h, size_t w) {
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "block has >= 2^31 pixels");
  if ((w & 3) != 0 || h < 2) return fail(c, WS_ERR_UNSUPPORTED, "the fast block form needs w % 4 == 0 and at least two rows (use ws_block_init / _relax / _resolve)");
  return WS_OK;
}

int ws_block_begin(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, uint8_t max_water_level,
                   const uint32_t *d_seeds_rc, size_t n_seeds, uint32_t first_colour, uint32_t *d_keys) {
  if (!c || !d_img || !d_keys || (n_seeds && !d_seeds_rc) || stride < w) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (max_water_level > WS_NORMAL_MAX) return fail(c, WS_ERR_MAX_TOO_HIGH, ws_strerror(WS_ERR_MAX_TOO_HIGH));
  if (max_water_level <= WS_ALWAYS_FILL) return fail(c, WS_ERR_MAX_TOO_LOW, ws_strerror(WS_ERR_MAX_TOO_LOW));
  if (n_seeds >= 0x7FFFFFFFull || (uint64_t)first_colour + n_seeds >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "colours must stay below 2^31");
  int rc = block_check(c, h, w);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const int ph = (int)h, pw = (int)w;
  const size_t n = h * w, nwords = (n + 31) / 32;
  if ((rc = ensure(c, c->stamps, std::max((size_t)tiles_of(pw) * tiles_of(ph), relax_tiles(ph, pw)) * 4 * 2 * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->seed_tab, nwords * 2 * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->refs, resolve_ref_capacity(ph, pw) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->tile_list, relax_list_words(ph, pw) * sizeof(uint32_t)))) return rc;
  uint32_t *tile_list = (uint32_t *)c->tile_list.p;
  uint32_t *flags = (uint32_t *)c->flags.p, *stamps = (uint32_t *)c->stamps.p;
  uint32_t *seed_mask = (uint32_t *)c->seed_tab.p, *word_base = seed_mask + nwords;
  c->have_keys = false;
  c->block_ready = false;
  stats_begin(c);
  if (!c->misc_clean) HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, FLAG_NERR * sizeof(uint32_t), c->stream));
  c->misc_clean = false;
  HIP_TRY(c, seed_tables(c->stream, d_seeds_rc, n_seeds, ph, pw, seed_mask, word_base, flags + FLAG_SEED_ERR, stamps,
                         relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC, nullptr, 0, first_colour - 1u));
  const PassFlags pf = make_pf(c);
  rc = pass_loop(c, flags, relax_tiles(ph, pw), &c->stats.relax_passes, [&](uint32_t pass) {
    Span sp(c, KC_RELAX);
    return relax_pass(c->stream, d_img, stride, d_keys, ph, pw, max_water_level, pass, stamps, pf, c->debug_max_iters, seed_mask, true, 0, true, false, tile_list);
  }, true, 5);
  if (rc) return rc;
  c->stats.launches_relax = c->stats.relax_passes;
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, FLAG_NERR * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_SEED_ERR]) return fail(c, WS_ERR_SEED_OOB, "seed outside the local plane");
  if (c->pinned[FLAG_NONSTRICT]) return fail(c, WS_ERR_UNSUPPORTED, "seed list not strictly increasing: use ws_block_init / _relax / _resolve");
  c->misc_clean = true;
  c->block_ready = true;
  c->block_h = h;
  c->block_w = w;
  return stats_end(c);
}

int ws_block_relax_halo(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, uint8_t max_water_level,
                        int halo_top, int halo_bottom, uint32_t *d_keys) {
  if (!c || !d_img || !d_keys || stride < w) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (!c->block_ready || c->block_h != h || c->block_w != w) return fail(c, WS_ERR_BAD_ARG, "ws_block_begin has not run for this block");
  const int halo = (halo_top ? 1 : 0) | (halo_bottom ? 2 : 0);
  if (!halo) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const int ph = (int)h, pw = (int)w;
  uint32_t *flags = (uint32_t *)c->flags.p, *stamps = (uint32_t *)c->stamps.p;
  const size_t ntiles = relax_tiles(ph, pw);
  stats_begin(c);
  HIP_TRY(c, hipMemsetAsync(stamps, 0, ntiles * 4 * 2 * sizeof(uint32_t), c->stream));
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_EDGE, 0, COUNTER_RING * FLAG_SLOT * sizeof(uint32_t), c->stream));
  uint32_t *tile_list = (uint32_t *)c->tile_list.p;      // sized by ws_block_begin
  constexpr uint32_t FIRST = 4;      // an even pass of the late kind: few tiles run (chunked launches, long-range scans)
  HIP_TRY(c, block_flag_border_tiles(c->stream, stamps, ph, pw, FIRST, halo));
  const PassFlags pf = make_pf(c);
  uint32_t last = 0;
  int rc = pass_loop(c, flags, ntiles, &last, [&](uint32_t pass) {
    Span sp(c, KC_RELAX);
    return relax_pass(c->stream, d_img, stride, d_keys, ph, pw, max_water_level, pass, stamps, pf, c->debug_max_iters, nullptr, false, 0, true, false, tile_list);
  }, true, 2, nullptr, nullptr, FIRST);
  if (rc) return rc;
  c->stats.relax_passes = last - FIRST;
  c->stats.launches_relax = c->stats.relax_passes;
  return stats_end(c);
}

int ws_block_resolve_local(ws_ctx *c, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w, int halo_top, int halo_bottom) {
  if (!c || !d_keys || !d_labels) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (!c->block_ready || c->block_h != h || c->block_w != w) return fail(c, WS_ERR_BAD_ARG, "ws_block_begin has not run for this block");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t nwords = (h * w + 31) / 32;
  uint32_t *flags = (uint32_t *)c->flags.p;
  uint32_t *seed_mask = (uint32_t *)c->seed_tab.p, *word_base = seed_mask + nwords;
  const int halo = (halo_top ? 1 : 0) | (halo_bottom ? 2 : 0);
  c->misc_clean = false;
  HIP_TRY(c, resolve_two_launch(c->stream, d_keys, d_labels, (int)h, (int)w, (uint32_t *)c->refs.p, c->debug_max_iters, seed_mask, word_base,
                                nullptr, nullptr, 0, flags + FLAG_OVERFLOW, flags + FLAG_SEED_ERR, halo));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_OVERFLOW]) return fail(c, WS_ERR_RING_OVERFLOW, "more than 2^24-1 flood rings inside one level");
  c->misc_clean = true;
  return WS_OK;
}

int ws_block_export_boundary(ws_ctx *c, const uint32_t *d_labels, size_t h, size_t w, int halo_top, int halo_bottom, size_t rank,
                             uint32_t *d_rows) {
  if (!c || !d_labels || !d_rows) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if ((halo_top && rank == 0) || h < (size_t)(1 + (halo_top ? 1 : 0) + (halo_bottom ? 1 : 0))) return fail(c, WS_ERR_BAD_ARG, "halo flags do not fit the block");
  if ((rank + 2) * 2 * w >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "boundary table index needs more than 31 bits");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_export_boundary(c->stream, d_labels, (int)h, (int)w, (halo_top ? 1 : 0) | (halo_bottom ? 2 : 0), (uint32_t)rank, d_rows));
  return WS_OK;
}

int ws_block_import_boundary(ws_ctx *c, const uint32_t *d_table, size_t world, size_t rank, uint32_t *d_labels, size_t h, size_t w,
                             int halo_top, int halo_bottom) {
  if (!c || !d_table || !d_labels || rank >= world) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (!c->block_ready || c->block_h != h || c->block_w != w) return fail(c, WS_ERR_BAD_ARG, "ws_block_resolve_local has not run for this block");
  if ((halo_top && rank == 0) || (halo_bottom && rank + 1 >= world)) return fail(c, WS_ERR_BAD_ARG, "halo flags do not fit the rank");
  if (world * 2 * w >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "boundary table index needs more than 31 bits");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->aux, world * 2 * w * sizeof(uint32_t)))) return rc;
  HIP_TRY(c, block_import_boundary(c->stream, d_table, (uint32_t)world, (uint32_t)rank, (uint32_t *)c->aux.p, d_labels, (int)h, (int)w,
                                   (halo_top ? 1 : 0) | (halo_bottom ? 2 : 0)));
  HIP_TRY(c, resolve_chase_again(c->stream, d_labels, (int)h, (int)w, (uint32_t *)c->refs.p));
  return WS_OK;
}

// ---- merging transform of a tiled field: final canonical labels (third row) ---------------------------------
//
// After the tiled segmenting transform (labels of the block final, halo rows included):
//   ws_block_merge_local    a union-find over ALL n_colours_total seed colours of the field in the caller's d_parent,
//                           the block's touching colours joined (ws_merge.hip, k_block_union_pixels)
//   ws_block_merge_export   (colour, root) of the block's boundary and halo rows: 4 * w pairs
//   all-gather of the pairs (distributed.py)
//   ws_block_merge_import   joins every gathered pair
//   ws_block_merge_relabel  d_out[p] = root(d_labels[p]): the smallest seed colour of the pixel's lake
int ws_block_merge_local(ws_ctx *c, const uint32_t *d_labels, size_t h, size_t w, size_t row0, size_t field_rows,
                         size_t n_colours_total, uint32_t *d_parent) {
  if (!c || !d_labels || !d_parent || row0 + h > field_rows) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || field_rows > 0x7FFFFFF0ull || n_colours_total >= 0x7FFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too large");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->uf_size, (n_colours_total + 1) * sizeof(uint32_t)))) return rc;      // uf_init's second array
  HIP_TRY(c, uf_init(c->stream, d_parent, (uint32_t *)c->uf_size.p, n_colours_total + 1));
  HIP_TRY(c, block_union_pixels(c->stream, d_labels, (int)h, (int)w, (int)row0, (int)field_rows, d_parent));
  return WS_OK;
}

int ws_block_merge_export(ws_ctx *c, const uint32_t *d_labels, size_t h, size_t w, uint32_t *d_parent, uint32_t *d_pairs) {
  if (!c || !d_labels || !d_parent || !d_pairs || h == 0) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_colour_roots(c->stream, d_labels, (int)h, (int)w, d_parent, (uint2 *)d_pairs));
  return WS_OK;
}

int ws_block_merge_import(ws_ctx *c, const uint32_t *d_pairs, size_t n_pairs, uint32_t *d_parent) {
  if (!c || !d_parent || (n_pairs && !d_pairs)) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  // (0, 0) pairs -- uncoloured boundary pixels -- join colour 0 with itself: nothing happens
  HIP_TRY(c, union_edges(c->stream, (const uint2 *)d_pairs, n_pairs, d_parent, nullptr, nullptr));
  return WS_OK;
}

int ws_block_merge_relabel(ws_ctx *c, const uint32_t *d_labels, size_t n, uint32_t *d_parent, size_t n_colours_total, uint32_t *d_out) {
  if (!c || !d_parent || (n && (!d_labels || !d_out))) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, relabel_final_u32(c->stream, d_labels, d_parent, n_colours_total + 1, d_out, n));
  return WS_OK;
}

// ---- merging (ws_merge.hip) ------------------------------------------------------------------

}  // extern "C"

namespace {

// mflags layout (u64 words): per-level histograms / cursors, lake cursor, per-level lake offsets
constexpr int MF_HIST_PX = 0;
constexpr int MF_HIST_ED = NLEVELS;
constexpr int MF_CUR_PX = 2 * NLEVELS;
constexpr int MF_CUR_ED = 3 * NLEVELS;
constexpr int MF_HOOKED = 5 * NLEVELS + 16;               // NLEVELS u32 counters (one per level: no memset between levels)
constexpr int MF_LAKE_COUNT = 6 * NLEVELS + 16;           // NLEVELS u64 per-level record counters
constexpr int MF_OFF_PX = 8 * NLEVELS + 24;               // NLEVELS + 1 bucket bounds of the arriving pixels (k_level_offsets)
constexpr int MF_OFF_ED = MF_OFF_PX + NLEVELS + 1;        // ... and of the crossing edges
constexpr int MF_WORDS = MF_OFF_ED + NLEVELS + 1;
constexpr uint32_t LIST_GROUP = 16;                       // levels per host copy of lake records (16 groups: kern_ev has 64)

// Segmenting result (stamps + colours) -> per-level buckets of arriving pixels and crossing edges, all on the device:
// histograms, their prefix sums (the bucket bounds, which only kernels ever read), scatter.  Nothing is read back, so
// the buffers are sized by what a plane can hold: one arrival per pixel, two crossing edges (right, down) per pixel.
int build_buckets(ws_ctx *c, const uint32_t *keys, const uint32_t *seg_labels, int ph, int pw) {
  int rc;
  const size_t n = (size_t)ph * pw;
  if ((rc = ensure(c, c->mflags, MF_WORDS * sizeof(uint64_t)))) return rc;
  if ((rc = ensure(c, c->px_items, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->edge_items, (n ? 2 * n : 1) * sizeof(uint2)))) return rc;
  u64c *mf = (u64c *)c->mflags.p;
  HIP_TRY(c, hipMemsetAsync(mf, 0, MF_WORDS * sizeof(uint64_t), c->stream));
  HIP_TRY(c, level_hist(c->stream, keys, seg_labels, ph, pw, mf + MF_HIST_PX, mf + MF_HIST_ED));
  HIP_TRY(c, level_offsets(c->stream, mf + MF_HIST_PX, mf + MF_HIST_ED, mf + MF_OFF_PX, mf + MF_OFF_ED, mf + MF_CUR_PX, mf + MF_CUR_ED));
  HIP_TRY(c, level_scatter(c->stream, keys, seg_labels, ph, pw, mf + MF_CUR_PX, mf + MF_CUR_ED,
                           (uint32_t *)c->px_items.p, (uint2 *)c->edge_items.p));
  return WS_OK;
}

int ensure_uf(ws_ctx *c, size_t n_colours) {
  int rc;
  if ((rc = ensure(c, c->uf_parent, n_colours * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->uf_size, n_colours * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->uf_hooked, n_colours * sizeof(uint32_t)))) return rc;
  if ((rc = ensure(c, c->uf_death, n_colours * sizeof(uint32_t)))) return rc;
  return WS_OK;
}

// workgroups of the per-level kernels: their buckets' sizes are only known on the device, so the grid follows the plane
// (an even spread would be pixels / 255 per level) and the kernels stride
unsigned level_grid(size_t n_px) { return (unsigned)std::min<size_t>(std::max<size_t>(n_px / (256 * 64), 8), 2048); }

// Levels [l0, l1) of the per-level driver shared by the merging hook and both transform_to_list flavours; launches only.
//   merging: union this level's crossing edges (lib.rs:1449-1466 in closed form)
//   want_sizes: keep per-lake areas (lib.rs:628-635)
//   per_level(l): called after level l is queued (device state current on c->stream)
//   fused (merging lists without a hook): the records of level l - 1 ride in the launch that joins level l's edges
//   (k_union_emit; ws_merge.hip) -- two launches per level instead of three; the caller emits the last level
struct FusedEmit {
  bool on = false;
  size_t n_colours = 0, cap = 0;
  uint64_t *lakes = nullptr;
};
template <class F>
int level_range(ws_ctx *c, uint32_t l0, uint32_t l1, bool merging, bool want_sizes, unsigned grid, F per_level, const FusedEmit &fe = FusedEmit()) {
  uint32_t *parent = (uint32_t *)c->uf_parent.p, *size = (uint32_t *)c->uf_size.p, *hooked = (uint32_t *)c->uf_hooked.p;
  u64c *mf = (u64c *)c->mflags.p;
  uint32_t *hooked_count = (uint32_t *)(mf + MF_HOOKED);
  const uint32_t *px_items = (const uint32_t *)c->px_items.p;
  const uint2 *edge_items = (const uint2 *)c->edge_items.p;
  for (uint32_t l = l0; l < l1; ++l) {
    if (fe.on) HIP_TRY(c, union_emit(c->stream, edge_items, mf + MF_OFF_ED + l, grid, parent, hooked, hooked_count + l, (uint32_t *)c->uf_death.p, l,
                                    size, fe.n_colours, fe.lakes, fe.cap, mf + MF_LAKE_COUNT));
    else if (merging) HIP_TRY(c, union_edges_ranged(c->stream, edge_items, mf + MF_OFF_ED + l, grid, parent, want_sizes ? hooked : nullptr, hooked_count + l));
    // areas of the nodes hooked in this level move to their roots, arriving pixels are counted: one launch
    if (want_sizes) HIP_TRY(c, fold_and_add_ranged(c->stream, merging ? hooked : nullptr, hooked_count + l, px_items, mf + MF_OFF_PX + l, grid, parent, size));
    int rc = per_level(l);
    if (rc) return rc;
  }
  return WS_OK;
}

// dev (nullable): the device-resident form (ws_transform_to_list_device) -- image and u32 seed pairs are already in HBM and
// the lake records stay there, in the caller's buffer; only the per-level offsets and uncoloured counts go to the host
struct DeviceLists {
  const uint8_t *d_img;
  const uint32_t *d_seeds_rc;
  ws_lake *d_lakes;
};
int merge_host(ws_ctx *c, bool merging, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
               size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels,
               ws_lake *lakes, size_t cap, size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured, const DeviceLists *dev = nullptr) {
  if (!c) return WS_ERR_BAD_ARG;
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = ph * pw;
  const bool want_list = n_lakes != nullptr;
  const uint8_t *d_img;
  size_t d_stride;
  const uint32_t *d_seeds;
  if ((rc = ensure(c, c->labels, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if (!dev && (rc = ensure(c, c->out64, (n ? n : 1) * sizeof(uint64_t)))) return rc;
  stats_begin(c);
  if (dev) {
    if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
    d_img = dev->d_img;
    d_stride = stride;
    if (opt->edge_correction && h * w == 0 && (rc = empty_image_block(c, &d_img, &d_stride))) return rc;
    if ((rc = shifted_seeds(c, dev->d_seeds_rc, n_seeds, opt, &d_seeds))) return rc;
  } else if ((rc = stage_inputs(c, img, h, w, stride, seeds_rc, n_seeds, opt, ph, pw, &d_img, &d_stride, &d_seeds))) return rc;
  uint32_t *seg = (uint32_t *)c->labels.p;
  uint64_t *d_out64 = (uint64_t *)c->out64.p;
  // the flood itself is the segmenting one (same coloured set, same arrival stamps: lib.rs:1394-1438 == 1704-1748)
  if ((rc = run_fused(c, d_img, d_stride, (int)ph, (int)pw, opt->max_water_level, d_seeds, n_seeds, seg, opt->edge_correction != 0))) return rc;
  const uint32_t *keys = (const uint32_t *)c->keys.p;
  // every buffer first, so that nothing moves once launches (or captured graphs) hold its address
  if ((rc = ensure_uf(c, n_seeds + 1))) return rc;
  if (want_list && !dev && (rc = ensure(c, c->lakes, (cap ? cap : 1) * 2 * sizeof(uint64_t)))) return rc;
  uint64_t *d_records = dev ? (uint64_t *)dev->d_lakes : (uint64_t *)c->lakes.p;      // (colour, area) pairs
  if ((rc = build_buckets(c, keys, seg, (int)ph, (int)pw))) return rc;
  uint32_t *parent = (uint32_t *)c->uf_parent.p;
  u64c *mf = (u64c *)c->mflags.p;
  HIP_TRY(c, uf_init(c->stream, parent, (uint32_t *)c->uf_size.p, n_seeds + 1));
  const uint8_t *himg = cb ? hook_image(c, img, h, w, stride, opt->edge_correction) : nullptr;
  if (cb) c->host64.resize(n ? n : 1);
  const uint32_t levels = (uint32_t)opt->max_water_level + 1;
  const unsigned grid = level_grid(n);

  // merging lists without a hook: level l's records are written by the launch that joins level l + 1's edges
  FusedEmit fe;
  fe.on = merging && want_list && !cb;
  fe.n_colours = n_seeds + 1; fe.cap = cap; fe.lakes = d_records;
  if (fe.on) HIP_TRY(c, hipMemsetAsync(c->uf_death.p, 0xFF, (n_seeds + 1) * sizeof(uint32_t), c->stream));      // every colour a root
  auto per_level = [&](uint32_t l) -> int {
    if (want_list && !fe.on)      // the kernel leaves this level's record count in its counter; offsets are prefix sums, taken on the host
      HIP_TRY(c, emit_lakes(c->stream, parent, (const uint32_t *)c->uf_size.p, n_seeds + 1, d_records, cap, mf + MF_LAKE_COUNT, l));
    if (cb) {
      if (merging) HIP_TRY(c, relabel_u64(c->stream, keys, seg, parent, d_out64, n, l));
      else HIP_TRY(c, snapshot_level(c->stream, keys, seg, d_out64, n, l));
      HIP_TRY(c, hipMemcpyAsync(c->host64.data(), d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      cb(user, (uint8_t)l, opt->max_water_level, himg, c->host64.data(), ph, pw);        // lib.rs:1510-1518
    }
    return WS_OK;
  };

  // The level loop is ~3 launches of a few microseconds per level and has no host decision in it: a call that repeats
  // the previous one's shape and buffers replays it as hipGraphs, one per group of LIST_GROUP levels (the groups' record
  // copies still overlap the later groups).  The second such call captures, later ones replay.
  ws_ctx::ListKey key;
  // (the fused-record mode follows from merging, want_list and cb == null)
  key.merging = merging; key.want_list = want_list; key.levels = levels; key.n_colours = n_seeds + 1; key.n = n; key.cap = cap;
  key.records = d_records;
  key.generation = c->buffer_generation;
  const bool graph_able = !cb && c->stream != nullptr && !c->graph_unusable && !c->profiling && n != 0;
  bool use_graphs = graph_able && key == c->list_seen_key;
  c->list_seen_key = graph_able ? key : ws_ctx::ListKey();
  if (!(use_graphs && key == c->list_graph_key)) {      // another shape: yesterday's graphs are of no use
    for (hipGraphExec_t &g : c->list_graphs) if (g) { (void)hipGraphExecDestroy(g); g = nullptr; }
    c->list_graph_key = use_graphs ? key : ws_ctx::ListKey();
  }
  if (tuning_env("WS_DEBUG_LIST"))
    std::fprintf(stderr, "[ws] merge_host: merging %d list %d cb %d levels %u colours %zu n %zu cap %zu gen %llu graph_able %d use_graphs %d fused %d\n",
                 (int)merging, (int)want_list, cb != nullptr, levels, n_seeds + 1, n, cap, (unsigned long long)key.generation, (int)graph_able, (int)use_graphs, (int)fe.on);
  for (uint32_t g0 = 0; g0 < levels; g0 += LIST_GROUP) {
    const uint32_t g1 = std::min(g0 + LIST_GROUP, levels), gi = g0 / LIST_GROUP;
    bool done = false;
    if (use_graphs) {
      if (!c->list_graphs[gi]) {
        hipGraph_t graph = nullptr;
        if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
          const int lrc = level_range(c, g0, g1, merging, want_list, grid, per_level, fe);
          const hipError_t e2 = hipStreamEndCapture(c->stream, &graph);
          if (lrc == WS_OK && e2 == hipSuccess && hipGraphInstantiate(&c->list_graphs[gi], graph, nullptr, nullptr, 0) != hipSuccess) c->list_graphs[gi] = nullptr;
          if (graph) (void)hipGraphDestroy(graph);
        }
        if (!c->list_graphs[gi]) {      // nothing ran: plain launches from here on, for good
          (void)hipGetLastError();
          c->graph_unusable = true;
          use_graphs = false;
        }
      }
      if (c->list_graphs[gi]) {
        HIP_TRY(c, hipGraphLaunch(c->list_graphs[gi], c->stream));
        c->stats.graph_launches++;
        done = true;
      }
    }
    if (!done && (rc = level_range(c, g0, g1, merging, want_list, grid, per_level, fe))) return rc;
    // a marker per group, so that the records of finished levels can travel to the host while later levels are computed
    if (want_list) HIP_TRY(c, hipEventRecord(c->kern_ev[gi], c->stream));
  }
  const uint32_t n_groups = (levels + LIST_GROUP - 1) / LIST_GROUP;
  if (fe.on) {      // the last level's records; and a marker behind them: in this mode a group's last level is complete one launch later
    HIP_TRY(c, emit_lakes(c->stream, parent, (const uint32_t *)c->uf_size.p, n_seeds + 1, d_records, cap, mf + MF_LAKE_COUNT, levels - 1,
                          (const uint32_t *)c->uf_death.p));
    HIP_TRY(c, hipEventRecord(c->kern_ev[n_groups], c->stream));
  }

  std::vector<uint64_t> bounds(2 * (NLEVELS + 1));
  if (want_list) {
    // All levels are queued.  Group by group: wait for the group's marker, read its offsets, copy its records
    // (155 MB at 1024^2: as long over PCIe as the levels take to compute, so the two are overlapped).
    offsets[0] = 0;
    size_t copied = 0;
    for (uint32_t g0 = 0; g0 < levels; g0 += LIST_GROUP) {
      const uint32_t g1 = std::min(g0 + LIST_GROUP, levels);
      // (fused records: group g's last level is written by group g + 1's first launch -- wait for that group's marker)
      HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->kern_ev[g0 / LIST_GROUP + (fe.on ? 1 : 0)], 0));
      HIP_TRY(c, hipMemcpyAsync(offsets + g0 + 1, mf + MF_LAKE_COUNT + g0, (g1 - g0) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->copy_stream));
      if (g0 == 0)      // the bucket bounds were final before the first level: they ride along with the first group
        HIP_TRY(c, hipMemcpyAsync(bounds.data(), mf + MF_OFF_PX, 2 * (NLEVELS + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->copy_stream));
      HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
      for (uint32_t l = g0; l < g1; ++l) offsets[l + 1] += offsets[l];      // counts -> offsets
      const size_t end = std::min<size_t>(offsets[g1], cap);
      if (!dev && end > copied) {
        HIP_TRY(c, hipMemcpyAsync(lakes + copied, (const ws_lake *)c->lakes.p + copied, (end - copied) * sizeof(ws_lake), hipMemcpyDeviceToHost, c->copy_stream));
        copied = end;
      }
    }
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    *n_lakes = offsets[levels];
    for (uint32_t l = 0; l < levels; ++l) uncoloured[l] = n - bounds[l + 1];                   // index 0 of lib.rs:630's vector
  } else {
    HIP_TRY(c, hipMemcpyAsync(bounds.data(), mf + MF_OFF_PX, 2 * (NLEVELS + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  }
  if (out_labels && n) {
    if (merging) HIP_TRY(c, relabel_u64(c->stream, keys, seg, parent, d_out64, n, opt->max_water_level));
    else HIP_TRY(c, widen_labels(c->stream, seg, d_out64, n));
    HIP_TRY(c, hipMemcpyAsync(out_labels, d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  }
  rc = stats_end(c);
  if (rc) return rc;
  if (merging)
    for (uint32_t l = 0; l < levels; ++l) c->stats.merge_levels += bounds[NLEVELS + 1 + l + 1] > bounds[NLEVELS + 1 + l] ? 1u : 0u;
  if (want_list && *n_lakes > cap) return fail(c, WS_ERR_CAPACITY, "lake buffer too small");
  return WS_OK;
}

}  // namespace

extern "C" {

// Final canonical labels only: one union pass over the whole image, no level buckets.
// A batch of equal-sized independent slices.  Fast path: the slices are stacked into ONE plane of S * h rows whose slice
// border rows are walls (they are image-border rows of their slices: never flooded), the seed lists are moved to stacked
// coordinates and the whole batch runs as a single transform -- one set of launches and one host round trip instead of
// S of each (8 x 4096^2: 2.0 -> ~1.4 ms; 16 x 1024^2: 2.2 ms -> ~0.3 ms).  Needs strictly increasing seed lists (the
// side-table form), w % 4 == 0, h * w % 128 == 0 and contiguous slices; anything else, and any error (so that the
// failing slice can be named), takes the slice-by-slice loop.
static int segment_batch_stacked(ws_ctx *c, const uint8_t *d_cube, size_t n_slices, size_t h, size_t w, size_t stride,
                                 const uint32_t *d_seeds_rc, const size_t *seed_offsets, const ws_options *opt,
                                 uint32_t *d_labels, bool *done) {
  *done = false;
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  static const bool off = tuning_env("WS_NO_BATCH_STACK") != nullptr || tuning_env("WS_NO_SEED_TABLES") != nullptr;      // A/B knobs for tools/
  const size_t plane = ph * pw;
  if (off || n_slices < 2 || pick_engine(opt) != WS_ENGINE_FUSED || !c->expect_sorted || (pw & 3) != 0 || plane == 0 ||
      plane % 128 != 0 || plane >= 0x40000000ull || stride != w || h * w == 0)
    return WS_OK;
  if (seed_offsets[n_slices] - seed_offsets[0] >= 0xFFFFFFFFull) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t max_px = c->batch_max_px;      // <= 2^31 - 1: the two-launch resolve indexes pixels with 31 bits
  const size_t per_group = std::max<size_t>(1, max_px / plane);
  std::vector<uint32_t> first;
  for (size_t k0 = 0; k0 < n_slices; k0 += per_group) {
    const size_t g = std::min(per_group, n_slices - k0);
    const size_t s0 = seed_offsets[k0], ns = seed_offsets[k0 + g] - s0;
    if (ns == 0) return WS_OK;
    stats_begin(c);
    const uint8_t *src = d_cube + k0 * h * stride;      // edge correction: the slices' rings of zeros are virtual
    const size_t src_stride = stride;
    first.resize(g + 1);
    for (size_t k = 0; k <= g; ++k) first[k] = (uint32_t)(seed_offsets[k0 + k] - s0);
    if ((rc = ensure(c, c->seed_stack, (ns * 2 + g + 1) * sizeof(uint32_t)))) return rc;
    uint32_t *stacked = (uint32_t *)c->seed_stack.p, *d_first = stacked + ns * 2;
    HIP_TRY(c, hipMemcpyAsync(d_first, first.data(), (g + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, stack_seeds(c->stream, d_seeds_rc + 2 * s0, ns, d_first, g, (int)ph, (int)pw, stacked, seed_shift_of(opt)));
    bool mispredicted = false;
    rc = run_fused_form(c, src, src_stride, (int)(g * ph), (int)pw, opt->max_water_level, stacked, ns, d_labels + k0 * plane, true,
                        &mispredicted, (int)ph, d_first, opt->edge_correction != 0);
    HIP_TRY(c, hipStreamSynchronize(c->stream));      // `first` is reused by the next group
    if (rc != WS_OK || mispredicted) {      // the loop repeats the work and names the slice
      (void)stats_end(c);                   // closes the span opened above; the loop's transforms keep their own statistics
      c->err.clear();
      return WS_OK;
    }
    c->have_keys = false;      // the stamps are those of a stack, not of an image
    if ((rc = stats_end(c))) return rc;
  }
  *done = true;
  return WS_OK;
}

int ws_segment_batch_device(ws_ctx *c, const uint8_t *d_cube, size_t n_slices, size_t h, size_t w, size_t stride,
                            size_t slice_stride, const uint32_t *d_seeds_rc, const size_t *seed_offsets,
                            const ws_options *opt, uint32_t *d_labels, size_t *failed_slice) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  if (failed_slice) *failed_slice = 0;
  if (n_slices && (!seed_offsets || !opt)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (n_slices > 1 && slice_stride < h * stride) return fail(c, WS_ERR_BAD_ARG, "slice_stride < h * row_stride");
  const size_t e = opt && opt->edge_correction ? 2 : 0, plane = (h + e) * (w + e);
  for (size_t k = 0; k < n_slices; ++k)
    if (seed_offsets[k + 1] < seed_offsets[k]) return fail(c, WS_ERR_BAD_ARG, "seed_offsets must not decrease");
  if (n_slices > 1 && slice_stride == h * stride && d_cube && d_seeds_rc && d_labels) {
    bool done = false;
    const int rc = segment_batch_stacked(c, d_cube, n_slices, h, w, stride, d_seeds_rc, seed_offsets, opt, d_labels, &done);
    if (rc != WS_OK) return rc;
    if (done) return WS_OK;
  }
  for (size_t k = 0; k < n_slices; ++k) {
    const int rc = ws_segment_device(c, d_cube + k * slice_stride, h, w, stride, d_seeds_rc + 2 * seed_offsets[k],
                                     seed_offsets[k + 1] - seed_offsets[k], opt, d_labels + k * plane);
    if (rc != WS_OK) { if (failed_slice) *failed_slice = k; return rc; }
  }
  return WS_OK;
}

// half: 0 the whole call; 1 ws_merge_device_begin (returns WS_INTERNAL_PENDING when the graph and the speculative unions
// have been queued and the host half is still to come); 2 ws_merge_device_end (that host half)
static int merge_device_body(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                             size_t n_seeds, const ws_options *opt, uint32_t *d_labels, int half) {
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  if ((!d_img && h * w) || (!d_seeds_rc && n_seeds) || (!d_labels && ph * pw)) return fail(c, WS_ERR_BAD_ARG, "null device pointer");
  if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = ph * pw;
  if ((rc = ensure(c, c->labels, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  stats_begin(c);
  const uint8_t *src = d_img;
  size_t src_stride = stride;
  const bool padded = opt->edge_correction != 0;
  if (padded && h * w == 0 && (rc = empty_image_block(c, &src, &src_stride))) return rc;
  const uint32_t *seeds;
  if ((rc = shifted_seeds(c, d_seeds_rc, n_seeds, opt, &seeds))) return rc;
  uint32_t *seg = (uint32_t *)c->labels.p;
  if ((rc = ensure(c, c->counts, std::max<size_t>(union_image_tiles((int)ph, (int)pw), 1) * sizeof(uint32_t)))) return rc;
  c->tile_min_out = (uint32_t *)c->counts.p;      // the resolve kernel classifies the tiles while it has them in registers
  c->tile_min_filled = false;
  if ((rc = ensure_uf(c, n_seeds + 1))) return rc;
  auto unions_and_relabel = [&](bool preclassified) -> int {
    HIP_TRY(c, uf_init(c->stream, (uint32_t *)c->uf_parent.p, (uint32_t *)c->uf_size.p, n_seeds + 1));
    Span sp(c, KC_OTHER);
    // at the final level a pixel is coloured exactly when its segmenting label is non-zero: no stamps needed
    HIP_TRY(c, union_image(c->stream, seg, seeds, n_seeds, (int)ph, (int)pw, (uint32_t *)c->uf_parent.p, (uint32_t *)c->counts.p,
                           preclassified, (uint32_t *)c->uf_size.p));      // (uf_init has just zeroed uf_size: the tile-root marks)
    HIP_TRY(c, relabel_final_u32(c->stream, seg, (uint32_t *)c->uf_parent.p, n_seeds + 1, d_labels, n, (uint32_t *)c->counts.p, (int)ph, (int)pw));
    return WS_OK;
  };
  // A call that replays the previous call's graph (run_fused_form: same buffers, sizes and seed count) queues its unions
  // and the relabel behind the graph BEFORE the host has looked at the graph's convergence word -- the host's wait and
  // look were ~18 us of idle GPU in the middle of every transform.  If the flood then turns out to need more passes (or
  // the seed tables were not valid), the unions ran on the previous call's labels and tile classes -- the same buffers,
  // valid colours of the same seed count -- and are simply done again after the real resolve.
  if (half != 2) {
    c->async_phase = ws_ctx::ASYNC_BEGIN;
    rc = run_fused(c, src, src_stride, (int)ph, (int)pw, opt->max_water_level, seeds, n_seeds, seg, padded);
  }
  bool speculated = false;
  if (half == 2 || (rc == WS_INTERNAL_PENDING && c->async_phase == ws_ctx::ASYNC_LAUNCHED)) {
    if (half != 2 && (rc = unions_and_relabel(true))) { c->async_phase = ws_ctx::ASYNC_NONE; c->tile_min_out = nullptr; return rc; }
    speculated = true;
    if (half == 1) return WS_INTERNAL_PENDING;      // (the context stays ASYNC_LAUNCHED, tile_min_out set: ws_merge_device_end)
    c->async_phase = ws_ctx::ASYNC_RESUME;      // the host half: waits for the graph's end event, reads its words, goes on if it must
    rc = run_fused(c, src, src_stride, (int)ph, (int)pw, opt->max_water_level, seeds, n_seeds, seg, padded);
  }
  c->async_phase = ws_ctx::ASYNC_NONE;
  c->tile_min_out = nullptr;
  if (rc) return rc == WS_INTERNAL_PENDING ? fail(c, WS_ERR_UNSUPPORTED, "internal: transform left pending") : rc;
  if (!(speculated && c->graph_sufficed) && (rc = unions_and_relabel(c->tile_min_filled))) return rc;
  c->stats.merge_levels = 1;
  return stats_end(c);
}

int ws_merge_device(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                    size_t n_seeds, const ws_options *opt, uint32_t *d_labels) {
  if (!c) return WS_ERR_BAD_ARG;
  if (c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "a transform begun with ws_*_device_begin has not been ended");
  return merge_device_body(c, d_img, h, w, stride, d_seeds_rc, n_seeds, opt, d_labels, 0);
}

// ws_merge_device in two halves, as ws_segment_device_begin / _end: what is left in flight is the replayed graph of the
// segmenting part AND the unions and the relabel queued behind it (see merge_device_body).
int ws_merge_device_begin(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const uint32_t *d_seeds_rc,
                          size_t n_seeds, const ws_options *opt, uint32_t *d_labels) {
  if (!c || !opt) return WS_ERR_BAD_ARG;
  if (c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "ws_merge_device_begin: the previous transform has not been ended");
  c->async_args = {d_img, h, w, stride, d_seeds_rc, n_seeds, *opt, d_labels};
  c->async_merge = true;
  const int rc = merge_device_body(c, d_img, h, w, stride, d_seeds_rc, n_seeds, opt, d_labels, 1);
  if (rc == WS_INTERNAL_PENDING && c->async_phase == ws_ctx::ASYNC_LAUNCHED) return WS_OK;
  c->async_phase = ws_ctx::ASYNC_DONE;      // ran whole (or failed): _end hands the status over
  c->async_rc = rc == WS_INTERNAL_PENDING ? (int)WS_ERR_UNSUPPORTED : rc;
  return WS_OK;
}

int ws_merge_device_end(ws_ctx *c) {
  if (!c) return WS_ERR_BAD_ARG;
  if (!c->async_merge) return fail(c, WS_ERR_BAD_ARG, "ws_merge_device_end without ws_merge_device_begin");
  if (c->async_phase == ws_ctx::ASYNC_DONE) { c->async_phase = ws_ctx::ASYNC_NONE; c->async_merge = false; return c->async_rc; }
  if (c->async_phase != ws_ctx::ASYNC_LAUNCHED) return fail(c, WS_ERR_BAD_ARG, "ws_merge_device_end without ws_merge_device_begin");
  const auto a = c->async_args;
  const int rc = merge_device_body(c, a.d_img, a.h, a.w, a.stride, a.d_seeds, a.n_seeds, &a.opt, a.d_labels, 2);
  c->async_phase = ws_ctx::ASYNC_NONE;
  c->async_merge = false;
  return rc == WS_INTERNAL_PENDING ? (int)WS_ERR_UNSUPPORTED : rc;
}

int ws_merge_with_hook(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                       size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  return merge_host(c, true, img, h, w, stride, seeds_rc, n_seeds, opt, cb, user, out_labels, nullptr, 0, nullptr, nullptr, nullptr);
}

int ws_transform_to_list_device(ws_ctx *c, int merging, const uint8_t *d_img, size_t h, size_t w, size_t stride,
                                const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt, ws_lake *d_lakes, size_t cap,
                                size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!n_lakes || !offsets || !uncoloured || (!d_lakes && cap) || (!d_img && h * w) || (!d_seeds_rc && n_seeds))
    return fail(c, WS_ERR_BAD_ARG, "null pointer");
  const DeviceLists dev{d_img, d_seeds_rc, d_lakes};
  return merge_host(c, merging != 0, nullptr, h, w, stride, nullptr, n_seeds, opt, nullptr, nullptr, nullptr, nullptr, cap, n_lakes, offsets,
                    uncoloured, &dev);
}

int ws_transform_to_list(ws_ctx *c, int merging, const uint8_t *img, size_t h, size_t w, size_t stride,
                         const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt, ws_lake *lakes, size_t cap,
                         size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!n_lakes || !offsets || !uncoloured || (!lakes && cap)) return fail(c, WS_ERR_BAD_ARG, "null output pointer");
  return merge_host(c, merging != 0, img, h, w, stride, seeds_rc, n_seeds, opt, nullptr, nullptr, nullptr, lakes, cap, n_lakes,
                    offsets, uncoloured);
}

}  // extern "C"
