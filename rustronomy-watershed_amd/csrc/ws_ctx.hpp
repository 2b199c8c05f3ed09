// ws_ctx.hpp -- what the host-side translation units of the C ABI share: the context, the flag-block layout, error and
// buffer helpers, the pass loop.  Kernels and their launch wrappers live in ws_common.hpp / ws_merge.hpp.
//
//   ws_ctx.hip        context life cycle, options, seeds (find_local_minima), pre-processor, staging helpers
//   ws_segment.hip    the segmenting drivers (lib.rs:1638-1808): fused and sweep engines, host / device / begin-end / batch
//   ws_lists.hip      the merging drivers and transform_to_list (lib.rs:1328-1522, 1551-1561)
//   ws_block_api.hip  row blocks of a tiled field, one call per step (the caller brings the collectives)
//   ws_tiled.hip      the same loop driven inside the library: several devices of one process, or one rank of an RCCL job
#pragma once

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

namespace wsapi {

struct DevBuf {
  void *p = nullptr;
  size_t cap = 0;
};

enum KClass { KC_RELAX = 0, KC_RESOLVE = 1, KC_SWEEP = 2, KC_OTHER = 3, KC_COUNT = 4 };

struct TimedSpan {
  hipEvent_t a, b;
  int cls;
};

}  // namespace wsapi

struct ws_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  bool profiling = false;
  std::string err;
  ws_stats stats{};

  wsapi::DevBuf img, keys, labels, labels2, stamps, flags, seeds, seeds64, out64, counts, aux, seed_stack, min_counts, min_nibbles;
  wsapi::DevBuf uf_parent, uf_size, uf_hooked, uf_death, uf_sd, alive, px_items, edge_items, mflags, lakes, refs, seed_tab, tile_list;
  uint32_t *pinned = nullptr;      // FLAG_WORDS words of pinned host memory: the host's mirror of the flag block
  uint32_t *pinned_dev = nullptr;  // the same words as the device sees them (nullptr: not mapped, copies only)
  hipEvent_t ring_ev[wsk::COUNTER_RING]{};   // flag slot copied to the host
  hipEvent_t kern_ev[wsk::COUNTER_RING]{};   // pass kernel finished
  hipStream_t copy_stream = nullptr;    // carries the per-pass flag read-backs
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t async_ev = nullptr;      // end of the graph a ws_segment_device_begin left in flight
  bool stats_no_wait = false;         // ws_segment_device_end: the stream may hold another context's work behind ours
  bool graph_sufficed = false;        // the last run_fused_form: replayed graph, at its fixpoint after the graph's passes, tables valid
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  std::vector<wsapi::TimedSpan> spans;
  std::vector<uint64_t> host64;    // hook staging
  uint32_t *hc_stage = nullptr;    // ws_hostcopy.hip: pinned staging slots of the chunked label copy (made on first use)
  hipEvent_t hc_ev[4]{};           // one per slot: its copy has landed
  std::vector<ws_ctx *> lanes;     // ws_segment_batch: internal contexts the slices of a host cube take turns on (owned; made on first use)
  int host_threads = 4;            // ... and the threads that widen the chunks (ws_ctx_set_host_threads; 0: one 8-byte copy instead)
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
  size_t minima_found_before = 0;           // ws_segment_minima: how many seeds the previous such transform found (0: none yet)
  int persistent_pass = 3;                  // long-range floods: the first same-grid pass as a persistent tile-queue launch: 0 never, 1 first come, 2 in flood order, 3 auto (ws_ctx_set_persistent_pass)
  size_t live_list_min = (size_t)1 << 20;   // fewest colours for which merging lists are written from the live-lake list (ws_ctx_set_live_list_min_colours)
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
    bool padded = false, from_minima = false;
    int persist_mode = 0;      // the captured passes depend on it (relax_pass)
    uint32_t max_level = 0;
    uint64_t generation = 0;      // of the context's own buffers (buffer_generation)
    bool operator==(const GraphKey &o) const {
      return generation == o.generation && from_minima == o.from_minima && persist_mode == o.persist_mode && img == o.img && seeds == o.seeds && labels == o.labels && slice_first == o.slice_first && tile_min == o.tile_min &&
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
    const void *keys = nullptr, *seg = nullptr;      // the planes the buckets are built from (the context's, or a caller's: the arrival form)
    uint64_t generation = 0;
    bool operator==(const ListKey &o) const {
      return generation == o.generation && generation != 0 && merging == o.merging && want_list == o.want_list && levels == o.levels &&
             n_colours == o.n_colours && n == o.n && cap == o.cap && records == o.records && keys == o.keys && seg == o.seg;
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

namespace wsapi {

using namespace wsk;

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

PassFlags make_pf(ws_ctx *c);
bool slot_nonzero(const uint32_t *slot);
int fail(ws_ctx *c, int code, const char *what, hipError_t e = hipSuccess);

#define HIP_TRY(ctx, call)                                              \
  do {                                                                  \
    hipError_t e_ = (call);                                             \
    if (e_ != hipSuccess) return ::wsapi::fail((ctx), e_ == hipErrorOutOfMemory ? WS_ERR_OOM : WS_ERR_HIP, #call, e_); \
  } while (0)

int ensure(ws_ctx *c, DevBuf &b, size_t bytes);
hipEvent_t next_event(ws_ctx *c);

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

void stats_begin(ws_ctx *c);
int stats_end(ws_ctx *c);
// ws_hostcopy.hip: the device's u32 labels into a host caller's u64 plane, 4 bytes a pixel over the bus, widened by host threads
// while the next chunks are in flight; returns with the copy complete (the stream has been waited for).
int labels_to_host_u64(ws_ctx *c, const uint32_t *d_labels, uint64_t *out, size_t n, hipStream_t on = nullptr,      // on: another stream of the context's (default: its own)
                       size_t row_len = 0, size_t out_pitch = 0);      // row_len != 0: rows of row_len words to rows of out_pitch words (chunked road only: host_copy_in_chunks)
void host_copy_release(ws_ctx *c);
bool host_copy_in_chunks(const ws_ctx *c, size_t n);      // whether labels_to_host_u64 widens on the host (else: on the device, into c->out64)
int check_plane(ws_ctx *c, size_t h, size_t w, size_t stride, const ws_options *opt, size_t *ph, size_t *pw);

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

// a context that holds a transform begun with ws_*_device_begin takes no other work until the matching _end
inline int refuse_if_in_flight(ws_ctx *c) {
  if (c && c->async_phase != ws_ctx::ASYNC_NONE) return fail(c, WS_ERR_BAD_ARG, "the context holds a transform begun with ws_*_device_begin: end it first");
  return WS_OK;
}

// ws_segment.hip
// (MinimaSeeds: the seeds are the image's own local minima, lib.rs:1178-1197 -- no list comes in; one goes out if asked for)
struct MinimaSeeds { uint32_t *d_list; size_t cap; size_t found; };
int run_fused_form(ws_ctx *c, const uint8_t *d_img, size_t stride, int ph, int pw, uint32_t max_level,
                   const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, bool tables, bool *mispredicted,
                   int slice_h = 0, const uint32_t *slice_first = nullptr, bool padded = false, MinimaSeeds *minima = nullptr);
int run_fused(ws_ctx *c, const uint8_t *d_img, size_t stride, int ph, int pw, uint32_t max_level,
              const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, bool padded = false);

inline int pick_engine(const ws_options *opt) {
  return opt->engine == WS_ENGINE_SWEEP ? WS_ENGINE_SWEEP : WS_ENGINE_FUSED;
}
// lib.rs:1675-1677 indexes the padded plane with the caller's coordinates (seed_shift 0); seed_shift 1 moves every seed
// by (+1, +1), onto the pixel it was found at.  Only meaningful with edge correction.
inline uint32_t seed_shift_of(const ws_options *opt) { return opt->edge_correction && opt->seed_shift ? 1u : 0u; }

// ws_ctx.hip: staging
int empty_image_block(ws_ctx *c, const uint8_t **d_img, size_t *d_stride, size_t h = 0, size_t w = 0);
int stage_inputs(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                 size_t n_seeds, const ws_options *opt, size_t ph, size_t pw, const uint8_t **d_img,
                 size_t *d_stride, const uint32_t **d_seeds);
int shifted_seeds(ws_ctx *c, const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt, const uint32_t **out);
const uint8_t *hook_image(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, bool edge);

}  // namespace wsapi
