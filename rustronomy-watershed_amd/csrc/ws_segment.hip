// ws_segment.hip -- the segmenting drivers: the reference's transform_with_hook body (lib.rs:1638-1808) restated as launch
// sequences on one HIP stream, in the fused form (all levels at once) and the literal sweep form.
#include "ws_ctx.hpp"

#include <atomic>
#include <mutex>
#include <system_error>
#include <thread>

namespace wsapi {

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
                   int slice_h, const uint32_t *slice_first, bool padded, MinimaSeeds *minima) {
  // minima: the seeds are the strict 8-neighbour maxima of d_img itself (what find_local_minima returns, in its order), and
  // the TABLES come from the same three launches that would have written that list -- the count per row, the scan, the
  // compaction -- instead of k_seed_tables' search through it: the README's call pair (lib.rs:73-86) without the 16-byte-
  // per-seed list in between (8192^2: 117 MB that a host caller would also have carried over PCIe, both ways).
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
  if (minima && (rc = ensure(c, c->min_counts, std::max<size_t>(minima_segments(ph, pw), 1) * sizeof(uint32_t)))) return rc;
  if (minima && (rc = ensure(c, c->min_nibbles, std::max<size_t>(minima_mask_bytes(ph, pw), 1)))) return rc;
  uint32_t *tile_list = (uint32_t *)c->tile_list.p;
  uint32_t *keys = (uint32_t *)c->keys.p;
  uint32_t *flags = (uint32_t *)c->flags.p;
  uint32_t *stamps = (uint32_t *)c->stamps.p;
  uint32_t *seed_mask = tables ? (uint32_t *)c->seed_tab.p : nullptr, *word_base = tables ? seed_mask + nwords : nullptr;
  c->have_keys = false;
  *mispredicted = false;
  c->graph_sufficed = false;

  // The late passes of a long-range flood as one queue launch in flood order (relax_pass, ws_ctx_set_persistent_pass)?  Auto:
  // when the seeds are sparse -- fewer than one per two of the 128 x 64 tiles, so that floods cross several tiles each.
  // 8192^2 smooth maps, correlation length 20 / 24 / 32 / 48 / 64 / 256 px (3.6 k / 1.7 k / 565 / 107 / 35 / 1 seeds), two draws
  // each: 6.1-6.3 -> 5.3-5.7, 6.6-7.5 -> 5.0-5.6, 8.7 -> 5.7-6.1, 8.4-8.9 -> 5.8-7.3, 6.9 -> 3.9, 3.6 -> 3.0 ms; at 16 px (8.6 k
  // seeds, one per tile) a draw, at 12 px and below (26 k seeds and more) the passes win (tools/exp_auto_check.py).
  // (seeds from the image's own minima: their number is only known afterwards -- the count of the context's previous such
  // transform stands in for it, a prediction like expect_sorted; a wrong one costs time, not labels)
  const size_t n_for_auto = minima ? c->minima_found_before : n_seeds;
  const int persist_mode = c->persistent_pass != 3 ? c->persistent_pass
                                                   : (n_for_auto >= 1 && n_for_auto <= relax_tiles(ph, pw) / 2 ? 2
                                                      : (n_for_auto >= 1 && n_for_auto <= relax_tiles(ph, pw) * 32 ? 4 : 0));      // (4: the passes, on the early schedule)
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
  key.from_minima = minima != nullptr;
  key.persist_mode = persist_mode;
  if (minima) { key.seeds = minima->d_list; key.n_seeds = minima->cap; }
  // the tables: from the list, or from the image
  auto make_tables = [&]() -> hipError_t {
    if (!minima)
      return seed_tables(c->stream, d_seeds, n_seeds, ph, pw, seed_mask, word_base, flags + FLAG_SEED_ERR, stamps,
                         relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC, slice_first, (size_t)slice_h * pw);
    uint32_t *counts = (uint32_t *)c->min_counts.p;
    uint8_t *nibbles = (uint8_t *)c->min_nibbles.p;
    hipError_t e = minima_count(c->stream, d_img, stride, ph, pw, counts, nibbles);
    if (e == hipSuccess)
      e = minima_write(c->stream, nibbles, ph, pw, counts, flags + FLAG_TOTAL, minima->d_list, minima->cap, seed_mask, word_base, stamps, relax_tiles(ph, pw) * 4 * 2, flags, FLAG_MISC);
    return e;
  };
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
    hipError_t e = make_tables();
    for (uint32_t pass = 0; pass < GRAPH_PASSES && e == hipSuccess; ++pass)
      e = relax_pass(c->stream, d_img, stride, keys, ph, pw, max_level, pass, stamps, gpf, c->debug_max_iters, seed_mask, true, slice_h, true, padded, tile_list, c->seam_min_px, persist_mode);
    const uint32_t last = GRAPH_PASSES - 1;
    if (e == hipSuccess)
      e = resolve_two_launch(c->stream, keys, d_labels, ph, pw, (uint32_t *)c->refs.p, c->debug_max_iters, seed_mask, word_base,
                             c->tile_min_out, edge_slot(flags, last), slice_h, flags + FLAG_OVERFLOW, flags + FLAG_SEED_ERR);
    // the read-backs: the lookahead pass's convergence slot and the error words
    if (e == hipSuccess && c->pinned_dev)
      e = words_to_host(c->stream, edge_slot(flags, last), FLAG_SLOT, c->pinned_dev + FLAG_EDGE + (last % COUNTER_RING) * FLAG_SLOT,
                        flags + FLAG_OVERFLOW, FLAG_NERR + 1, c->pinned_dev + FLAG_OVERFLOW);      // (+ 1: FLAG_TOTAL, the minima's count)
    if (e == hipSuccess && !c->pinned_dev)
      e = hipMemcpyAsync(&c->pinned[FLAG_EDGE + (last % COUNTER_RING) * FLAG_SLOT], edge_slot(flags, last), FLAG_SLOT * sizeof(uint32_t),
                         hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess && !c->pinned_dev)
      e = hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, (FLAG_NERR + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream);
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
      HIP_TRY(c, make_tables());
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
                      tables ? seed_mask : d_labels, tables, slice_h, two_launch, padded, tile_list, c->seam_min_px, persist_mode);
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
    HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, (FLAG_NERR + 1) * sizeof(uint32_t),
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
  if (minima) { minima->found = c->pinned[FLAG_TOTAL]; c->minima_found_before = minima->found; }
  if (c->pinned[FLAG_SEED_ERR]) return fail(c, WS_ERR_SEED_OOB, "seed outside the label plane (the reference panics: lib.rs:1676)");
  if (!minima) c->expect_sorted = c->pinned[FLAG_NONSTRICT] == 0;
  if (tables && !minima && !c->expect_sorted) {      // the tables describe some other list: nothing computed from them counts
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
              const uint32_t *d_seeds, size_t n_seeds, uint32_t *d_labels, bool padded) {
  // the side-table form needs nibble-aligned patch rows (W % 4 == 0) and the two-launch resolve
  static const bool no_tables = tuning_env("WS_NO_SEED_TABLES") != nullptr;      // A/B knob for tools/
  const bool can_tables = !no_tables && (pw & 3) == 0 && (size_t)ph * pw < 0x80000000ull && n_seeds > 0;
  bool mispredicted = false;
  int rc = run_fused_form(c, d_img, stride, ph, pw, max_level, d_seeds, n_seeds, d_labels, can_tables && c->expect_sorted, &mispredicted, 0, nullptr, padded);
  if (rc == WS_OK && mispredicted) rc = run_fused_form(c, d_img, stride, ph, pw, max_level, d_seeds, n_seeds, d_labels, false, &mispredicted, 0, nullptr, padded);
  return rc;
}

}  // namespace wsapi

using namespace wsapi;

namespace {

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

int segment_host(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                 size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels, uint32_t *out_labels_u32 = nullptr) {
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
  if (cb && (rc = ensure(c, c->out64, (n ? n : 1) * sizeof(uint64_t)))) return rc;      // the hook's planes are widened on the device
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
        if (host_copy_in_chunks(c, n)) {      // the level's plane as u32 (in the u64 buffer, which that path leaves alone), widened by host threads
          HIP_TRY(c, snapshot_level_u32(c->stream, (const uint32_t *)c->keys.p, d_labels, (uint32_t *)d_out64, n, lvl));
          if ((rc = labels_to_host_u64(c, (const uint32_t *)d_out64, c->host64.data(), n))) return rc;
        } else {
          HIP_TRY(c, snapshot_level(c->stream, (const uint32_t *)c->keys.p, d_labels, d_out64, n, lvl));
          HIP_TRY(c, hipMemcpyAsync(c->host64.data(), d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
          HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
        cb(user, (uint8_t)lvl, opt->max_water_level, himg, c->host64.data(), ph, pw);
      }
    }
  }
  if (out_labels && n) {
    Span sp(c, KC_OTHER);
    if ((rc = labels_to_host_u64(c, d_labels, out_labels, n))) return rc;
  }
  if (out_labels_u32 && n)      // ws_segment_u32: the device's own 4-byte labels, half the bytes over PCIe
    HIP_TRY(c, hipMemcpyAsync(out_labels_u32, d_labels, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  return stats_end(c);
}

}  // namespace

extern "C" {

// ---- segmenting ---------------------------------------------------------------------------

int ws_segment(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
               size_t n_seeds, const ws_options *opt, uint64_t *out_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!out_labels) return fail(c, WS_ERR_BAD_ARG, "out_labels is null");
  return segment_host(c, img, h, w, stride, seeds_rc, n_seeds, opt, nullptr, nullptr, out_labels);
}

int ws_segment_u32(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                   size_t n_seeds, const ws_options *opt, uint32_t *out_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!out_labels) return fail(c, WS_ERR_BAD_ARG, "out_labels is null");
  return segment_host(c, img, h, w, stride, seeds_rc, n_seeds, opt, nullptr, nullptr, nullptr, out_labels);
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
  if (padded && h * w == 0 && (rc = empty_image_block(c, &src, &src_stride, h, w))) return rc;
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

// ---- the README's call pair as one call (lib.rs:73-86): seeds = find_local_minima(img), labels = transform(img, seeds) -----------
// Fast form: fused engine, no edge correction, w % 32 == 0 -- the seed side tables come straight out of the minima kernels
// (run_fused_form, MinimaSeeds) and the list is only written when the caller wants it.  Anything else runs the two calls.
static int segment_minima_device_body(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const ws_options *opt,
                                      uint32_t *d_labels, uint32_t *d_seeds_out, size_t cap, size_t *n_found) {
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  if (!n_found || (!d_img && h * w) || (!d_labels && ph * pw) || (!d_seeds_out && cap)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  *n_found = 0;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool fast = pick_engine(opt) == WS_ENGINE_FUSED && !opt->edge_correction && h >= 3 && w >= 3 && (w & 31) == 0 && h * w < 0x80000000ull;
  if (fast) {
    stats_begin(c);
    MinimaSeeds ms{d_seeds_out, cap, 0};
    bool mispredicted = false;
    rc = run_fused_form(c, d_img, stride, (int)h, (int)w, opt->max_water_level, nullptr, 0, d_labels, true, &mispredicted, 0, nullptr, false, &ms);
    if (rc) return rc;
    *n_found = ms.found;
    if ((rc = stats_end(c))) return rc;
    // (an image without a single minimum: no seed, nothing is ever coloured -- the tables say so and the labels are zero)
    return d_seeds_out && ms.found > cap ? fail(c, WS_ERR_CAPACITY, "seed buffer too small (the labels are complete)") : WS_OK;
  }
  // the two calls, through a list of the context's own when the caller's cannot hold it
  const size_t bound = h >= 3 && w >= 3 ? ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1) : 0;      // at most one strict maximum per 2 x 2 block
  uint32_t *list = d_seeds_out;
  size_t lcap = cap;
  if (cap < bound) {
    if ((rc = ensure(c, c->seed_stack, std::max<size_t>(bound, 1) * 2 * sizeof(uint32_t)))) return rc;
    list = (uint32_t *)c->seed_stack.p;
    lcap = bound;
  }
  size_t found = 0;
  if ((rc = ws_find_local_minima_device(c, d_img, h, w, stride, list, lcap, &found))) return rc;
  *n_found = found;
  if ((rc = segment_device_body(c, d_img, h, w, stride, list, found, opt, d_labels))) return rc;
  if (list != d_seeds_out && cap && found)
    HIP_TRY(c, hipMemcpyAsync(d_seeds_out, list, std::min(found, cap) * 2 * sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
  if (list != d_seeds_out && cap) HIP_TRY(c, hipStreamSynchronize(c->stream));
  return d_seeds_out && found > cap ? fail(c, WS_ERR_CAPACITY, "seed buffer too small (the labels are complete)") : WS_OK;
}

int ws_segment_minima_device(ws_ctx *c, const uint8_t *d_img, size_t h, size_t w, size_t stride, const ws_options *opt,
                             uint32_t *d_labels, uint32_t *d_seeds_rc, size_t cap, size_t *n_seeds) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  return segment_minima_device_body(c, d_img, h, w, stride, opt, d_labels, d_seeds_rc, cap, n_seeds);
}

static int segment_minima_host(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const ws_options *opt,
                               uint64_t *out64, uint32_t *out32, uint64_t *seeds_rc, size_t cap, size_t *n_seeds) {
  if (!c) return WS_ERR_BAD_ARG;
  if (!n_seeds || (!img && h * w) || (!seeds_rc && cap) || (!out64 && !out32)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = ph * pw;
  if ((rc = ensure(c, c->img, std::max<size_t>(h * w, 1)))) return rc;
  if ((rc = ensure(c, c->labels, std::max<size_t>(n, 1) * sizeof(uint32_t)))) return rc;
  const size_t bound = h >= 3 && w >= 3 ? ((h - 1) / 2 + 1) * ((w - 1) / 2 + 1) : 0;
  const size_t dcap = std::min(cap, bound);
  if (dcap && (rc = ensure(c, c->seeds64, dcap * 2 * sizeof(uint32_t)))) return rc;
  if (h * w) HIP_TRY(c, hipMemcpy2DAsync(c->img.p, w, img, stride, w, h, hipMemcpyHostToDevice, c->stream));
  uint32_t *d_list = dcap ? (uint32_t *)c->seeds64.p : nullptr;
  rc = segment_minima_device_body(c, (const uint8_t *)c->img.p, h, w, w, opt, (uint32_t *)c->labels.p, d_list, dcap, n_seeds);
  const bool short_list = rc == WS_ERR_CAPACITY;
  if (rc != WS_OK && !short_list) return rc;
  const size_t got = std::min(*n_seeds, dcap);
  // (pairs and labels alike: u32 over the bus, widened by host threads when there are 2^21 words or more -- ws_hostcopy.hip;
  // each call returns with its copy complete, so the small-plane path's one staging buffer serves both)
  if (got && (rc = labels_to_host_u64(c, d_list, seeds_rc, got * 2))) return rc;
  if (out64 && n && (rc = labels_to_host_u64(c, (const uint32_t *)c->labels.p, out64, n))) return rc;
  if (out32 && n) HIP_TRY(c, hipMemcpyAsync(out32, c->labels.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (seeds_rc && (short_list || *n_seeds > cap)) return fail(c, WS_ERR_CAPACITY, "seed buffer too small (the labels are complete)");
  return WS_OK;
}

int ws_segment_minima(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const ws_options *opt, uint64_t *out_labels,
                      uint64_t *seeds_rc, size_t cap, size_t *n_seeds) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  return segment_minima_host(c, img, h, w, stride, opt, out_labels, nullptr, seeds_rc, cap, n_seeds);
}

int ws_segment_minima_u32(ws_ctx *c, const uint8_t *img, size_t h, size_t w, size_t stride, const ws_options *opt, uint32_t *out_labels,
                          uint64_t *seeds_rc, size_t cap, size_t *n_seeds) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  return segment_minima_host(c, img, h, w, stride, opt, nullptr, out_labels, seeds_rc, cap, n_seeds);
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

// A cube of independent slices in HOST memory (tests/integration.rs:267,356: the reference walks the slices of a CGPS cube one
// call of transform() after the other).  One slice's call is upload, transform, label copy in a row: the bus idles while the
// transform runs and carries one direction at a time.  Here the slices take turns on BATCH_LANES internal contexts, each with a
// host thread and a stream of its own: one slice's image goes up while another's labels come down and a third transforms.
// Every slice is exactly ws_segment_minima (seeds_rc == NULL) or ws_segment of that slice.
constexpr int BATCH_LANES = 4;      // (16 x 4096^2, ms per slice: 1 lane 2.3 = the loop, 2: 1.58, 3: 1.50, 4: 1.44, 6: 1.36-1.45; the label copy alone: 1.17)

int ws_segment_batch(ws_ctx *c, const uint8_t *cube, size_t n_slices, size_t h, size_t w, size_t stride, size_t slice_stride,
                     const uint64_t *seeds_rc, const size_t *seed_offsets, const ws_options *opt, uint64_t *out_labels,
                     size_t *n_seeds, size_t *failed_slice) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  if (failed_slice) *failed_slice = 0;
  if (n_slices == 0) return WS_OK;
  if (!opt || !out_labels || (!cube && h * w) || (seeds_rc && !seed_offsets)) return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (n_slices > 1 && slice_stride < h * stride) return fail(c, WS_ERR_BAD_ARG, "slice_stride < h * row_stride");
  size_t ph, pw;
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  if (seeds_rc)
    for (size_t k = 0; k < n_slices; ++k)
      if (seed_offsets[k + 1] < seed_offsets[k]) return fail(c, WS_ERR_BAD_ARG, "seed_offsets must not decrease");
  const size_t plane = ph * pw;
  static const int max_lanes = [] { const char *e = tuning_env("WS_BATCH_LANES"); return e ? std::max(1, std::min(8, atoi(e))) : BATCH_LANES; }();      // A/B knob, tools/ only
  const int lanes = (int)std::min<size_t>((size_t)max_lanes, n_slices);
  while ((int)c->lanes.size() < lanes) {
    ws_ctx *lane = nullptr;
    if ((rc = ws_ctx_create(c->device, &lane))) return fail(c, rc, "a lane context of ws_segment_batch could not be created");
    c->lanes.push_back(lane);
  }
  for (int t = 0; t < lanes; ++t) {      // the lanes follow the context's settings
    ws_ctx *lane = c->lanes[t];
    lane->host_threads = c->host_threads;
    lane->seam_min_px = c->seam_min_px;
    if (lane->persistent_pass != c->persistent_pass) { lane->persistent_pass = c->persistent_pass; ++lane->buffer_generation; }
  }
  auto one_slice = [&](ws_ctx *lane, size_t k) -> int {
    const uint8_t *img = cube + k * slice_stride;
    uint64_t *out = out_labels + k * plane;
    if (seeds_rc) {
      const size_t ns = seed_offsets[k + 1] - seed_offsets[k];
      if (n_seeds) n_seeds[k] = ns;
      return segment_host(lane, img, h, w, stride, seeds_rc + 2 * seed_offsets[k], ns, opt, nullptr, nullptr, out);
    }
    size_t found = 0;
    const int r = segment_minima_host(lane, img, h, w, stride, opt, out, nullptr, nullptr, 0, &found);
    if (n_seeds) n_seeds[k] = found;
    return r;
  };
  std::mutex m;
  int first_rc = WS_OK;
  size_t first_slice = 0;
  std::string first_msg;
  std::atomic<size_t> stop_at{n_slices};      // the lowest slice that failed so far: slices below it still run, so that the
  auto run_lane = [&](int t) {                // slice reported is the lowest failing one whatever the lanes' pace
    ws_ctx *lane = c->lanes[t];
    for (size_t k = (size_t)t; k < stop_at.load(std::memory_order_relaxed); k += (size_t)lanes) {
      const int r = one_slice(lane, k);
      if (r != WS_OK) {
        std::lock_guard<std::mutex> lk(m);
        if (first_rc == WS_OK || k < first_slice) { first_rc = r; first_slice = k; first_msg = ws_last_error(lane); }
        if (k < stop_at.load()) stop_at.store(k);
        return;
      }
    }
  };
  std::vector<std::thread> pool;
  bool threads_ok = true;
  try {
    for (int t = 1; t < lanes; ++t) pool.emplace_back(run_lane, t);
  } catch (const std::system_error &) {
    threads_ok = false;      // the process may not start threads: the lanes that have none are run below, one after the other
  }
  run_lane(0);
  const int started = (int)pool.size() + 1;
  for (auto &th : pool) th.join();
  if (!threads_ok)
    for (int t = started; t < lanes; ++t) run_lane(t);
  if (first_rc != WS_OK) {
    if (failed_slice) *failed_slice = first_slice;
    c->err = "slice " + std::to_string(first_slice) + ": " + first_msg;
    return first_rc;
  }
  return WS_OK;
}

}  // extern "C"
