// ws_lists.hip -- the merging drivers (lib.rs:1328-1522) and transform_to_list (lib.rs:1551-1561, 1837-1847): per-level
// unions over seed colours, lake areas and sparse lake records (kernels: ws_merge.hip).
#include "ws_ctx.hpp"

using namespace wsapi;

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
// histograms, their prefix sums (the bucket bounds, which only kernels ever read), scatter.  Workspace: 4 B per arriving
// pixel + 8 B per crossing edge, at most 20 B per pixel of the plane (one arrival per pixel, two crossing edges -- right,
// down -- per pixel; a random field is close to that).  A context whose buffers already hold the worst case reads
// nothing back; otherwise the two totals (16 bytes) are read once and the buffers sized by them, so that a sparse or
// partly flooded plane near the 2^32-pixel limit does not ask for 80 GB it will not use.
int build_buckets(ws_ctx *c, const uint32_t *keys, const uint32_t *seg_labels, int ph, int pw) {
  int rc;
  const size_t n = (size_t)ph * pw;
  if ((rc = ensure(c, c->mflags, MF_WORDS * sizeof(uint64_t)))) return rc;
  u64c *mf = (u64c *)c->mflags.p;
  HIP_TRY(c, hipMemsetAsync(mf, 0, MF_WORDS * sizeof(uint64_t), c->stream));
  HIP_TRY(c, level_hist(c->stream, keys, seg_labels, ph, pw, mf + MF_HIST_PX, mf + MF_HIST_ED));
  HIP_TRY(c, level_offsets(c->stream, mf + MF_HIST_PX, mf + MF_HIST_ED, mf + MF_OFF_PX, mf + MF_OFF_ED, mf + MF_CUR_PX, mf + MF_CUR_ED));
  size_t need_px = (n ? n : 1) * sizeof(uint32_t), need_ed = (n ? 2 * n : 1) * sizeof(uint2);
  if (c->px_items.cap < need_px || c->edge_items.cap < need_ed) {
    unsigned long long totals[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(&totals[0], mf + MF_OFF_PX + NLEVELS, sizeof(u64c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(&totals[1], mf + MF_OFF_ED + NLEVELS, sizeof(u64c), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    need_px = std::min<size_t>(need_px, std::max<size_t>(totals[0], 1) * sizeof(uint32_t));
    need_ed = std::min<size_t>(need_ed, std::max<size_t>(totals[1], 1) * sizeof(uint2));
  }
  if ((rc = ensure(c, c->px_items, need_px))) return rc;
  if ((rc = ensure(c, c->edge_items, need_ed))) return rc;
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
  bool live = false;           // many colours: records from the list of LIVE lakes (ws_merge.hip) instead of a look at every colour per level
  unsigned emit_grid = 0;      // workgroups of the live-list walk
};
template <class F>
int level_range(ws_ctx *c, uint32_t l0, uint32_t l1, bool merging, bool want_sizes, unsigned grid, F per_level, const FusedEmit &fe = FusedEmit()) {
  uint32_t *parent = (uint32_t *)c->uf_parent.p, *size = (uint32_t *)c->uf_size.p, *hooked = (uint32_t *)c->uf_hooked.p;
  u64c *mf = (u64c *)c->mflags.p;
  uint32_t *hooked_count = (uint32_t *)(mf + MF_HOOKED);
  const uint32_t *px_items = (const uint32_t *)c->px_items.p;
  const uint2 *edge_items = (const uint2 *)c->edge_items.p;
  for (uint32_t l = l0; l < l1; ++l) {
    if (fe.on && !fe.live) {
      HIP_TRY(c, union_emit(c->stream, edge_items, mf + MF_OFF_ED + l, grid, parent, hooked, hooked_count + l, (uint32_t *)c->uf_death.p, l,
                            size, fe.n_colours, fe.lakes, fe.cap, mf + MF_LAKE_COUNT));
      HIP_TRY(c, fold_and_add_ranged(c->stream, hooked, hooked_count + l, px_items, mf + MF_OFF_PX + l, grid, parent, size));
      int rc = per_level(l);
      if (rc) return rc;
      continue;
    }
    if (fe.on) {
      // lists without a hook, many colours: areas and death levels live side by side in uf_sd, level l - 1's records are found among level
      // l - 2's live lakes, and the arrivals of a level are added up per wave and workgroup before they reach a lake's counter
      static const bool split = tuning_env("WS_TOLIST_SPLIT") != nullptr;      // A/B knob for tools/: the two jobs as launches of their own
      HIP_TRY(c, union_emit_alive(c->stream, edge_items, mf + MF_OFF_ED + l, grid, parent, hooked, hooked_count + l, (uint2 *)c->uf_sd.p, l,
                                  fe.n_colours, (uint32_t *)c->alive.p, split ? 0u : fe.emit_grid, fe.lakes, fe.cap, mf + MF_LAKE_COUNT));
      if (split && l > 0)
        HIP_TRY(c, emit_alive(c->stream, (const uint2 *)c->uf_sd.p, fe.n_colours, (uint32_t *)c->alive.p, fe.emit_grid, fe.lakes, fe.cap, mf + MF_LAKE_COUNT, l - 1));
      HIP_TRY(c, fold_and_add_sd(c->stream, hooked, hooked_count + l, px_items, mf + MF_OFF_PX + l, grid, parent, (uint2 *)c->uf_sd.p));
      int rc = per_level(l);
      if (rc) return rc;
      continue;
    }
    if (merging) HIP_TRY(c, union_edges_ranged(c->stream, edge_items, mf + MF_OFF_ED + l, grid, parent, want_sizes ? hooked : nullptr, hooked_count + l));
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
  // the arrival form (ws_lists_from_arrival_device): the segmenting transform has been run elsewhere -- its stamps and labels
  // are all the per-level paths read; no image, no seed list, h x w is the plane as it stands
  const uint32_t *d_keys = nullptr, *d_seg = nullptr;
};
int merge_host(ws_ctx *c, bool merging, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
               size_t n_seeds, const ws_options *opt, ws_level_cb cb, void *user, uint64_t *out_labels,
               ws_lake *lakes, size_t cap, size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured, const DeviceLists *dev = nullptr) {
  if (!c) return WS_ERR_BAD_ARG;
  size_t ph, pw;
  const bool from_arrival = dev && dev->d_keys;
  ws_options plain;
  if (from_arrival) { plain = *opt; plain.edge_correction = 0; opt = &plain; }      // the plane as it stands
  int rc = check_plane(c, h, w, stride, opt, &ph, &pw);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t n = ph * pw;
  const bool want_list = n_lakes != nullptr;
  const uint8_t *d_img = nullptr;
  size_t d_stride = 0;
  const uint32_t *d_seeds = nullptr;
  if (!from_arrival && (rc = ensure(c, c->labels, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if (!dev && (rc = ensure(c, c->out64, (n ? n : 1) * sizeof(uint64_t)))) return rc;
  stats_begin(c);
  if (from_arrival) {
    if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
  } else if (dev) {
    if (n_seeds >= 0xFFFFFFFFull) return fail(c, WS_ERR_TOO_LARGE, "too many seeds");
    d_img = dev->d_img;
    d_stride = stride;
    if (opt->edge_correction && h * w == 0 && (rc = empty_image_block(c, &d_img, &d_stride, h, w))) return rc;
    if ((rc = shifted_seeds(c, dev->d_seeds_rc, n_seeds, opt, &d_seeds))) return rc;
  } else if ((rc = stage_inputs(c, img, h, w, stride, seeds_rc, n_seeds, opt, ph, pw, &d_img, &d_stride, &d_seeds))) return rc;
  const uint32_t *seg = from_arrival ? dev->d_seg : (const uint32_t *)c->labels.p;
  uint64_t *d_out64 = (uint64_t *)c->out64.p;
  // the flood itself is the segmenting one (same coloured set, same arrival stamps: lib.rs:1394-1438 == 1704-1748)
  if (!from_arrival && (rc = run_fused(c, d_img, d_stride, (int)ph, (int)pw, opt->max_water_level, d_seeds, n_seeds, (uint32_t *)c->labels.p, opt->edge_correction != 0))) return rc;
  const uint32_t *keys = from_arrival ? dev->d_keys : (const uint32_t *)c->keys.p;
  // every buffer first, so that nothing moves once launches (or captured graphs) hold its address
  if ((rc = ensure_uf(c, n_seeds + 1))) return rc;
  // Planes with a million colours and more (4096^2 random fields on) write their lake records from the list of the lakes
  // still alive, not from a look at every colour at every level (8192^2: 66.8 -> 20.5 ms); below that the per-level
  // launches are latency-bound either way and the older, shorter kernels win (1024^2, the core_bench shape: 3.8 against 5.0 ms).
  // (ws_ctx_set_live_list_min_colours: tests lower the threshold to cover the form on small planes)
  const bool live_lists = merging && want_list && !cb && n_seeds >= c->live_list_min;
  if (live_lists) {
    if ((rc = ensure(c, c->uf_sd, (n_seeds + 1) * sizeof(uint2)))) return rc;
    if ((rc = ensure(c, c->alive, 2 * alive_list_words(n_seeds + 1) * sizeof(uint32_t)))) return rc;
  }
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
  fe.live = live_lists;
  if (fe.on && fe.live) {
    fe.emit_grid = (unsigned)std::min<size_t>(std::max<size_t>((n_seeds + 4095) / 4096, 1), 1024);
    HIP_TRY(c, sd_init(c->stream, (uint2 *)c->uf_sd.p, n_seeds + 1));      // no pixels yet, every colour a root
  } else if (fe.on) {
    HIP_TRY(c, hipMemsetAsync(c->uf_death.p, 0xFF, (n_seeds + 1) * sizeof(uint32_t), c->stream));      // every colour a root
  }
  auto per_level = [&](uint32_t l) -> int {
    if (want_list && !fe.on)      // the kernel leaves this level's record count in its counter; offsets are prefix sums, taken on the host
      HIP_TRY(c, emit_lakes(c->stream, parent, (const uint32_t *)c->uf_size.p, n_seeds + 1, d_records, cap, mf + MF_LAKE_COUNT, l));
    if (cb) {
      if (host_copy_in_chunks(c, n)) {      // the level's plane as u32 (in the u64 buffer, which that path leaves alone), widened by host threads
        if (merging) HIP_TRY(c, relabel_u32(c->stream, keys, seg, parent, (uint32_t *)d_out64, n, l));
        else HIP_TRY(c, snapshot_level_u32(c->stream, keys, seg, (uint32_t *)d_out64, n, l));
        if (int rc_copy = labels_to_host_u64(c, (const uint32_t *)d_out64, c->host64.data(), n)) return rc_copy;
      } else {
        if (merging) HIP_TRY(c, relabel_u64(c->stream, keys, seg, parent, d_out64, n, l));
        else HIP_TRY(c, snapshot_level(c->stream, keys, seg, d_out64, n, l));
        HIP_TRY(c, hipMemcpyAsync(c->host64.data(), d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
      }
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
  key.keys = keys; key.seg = seg;
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
    if (fe.live)
      HIP_TRY(c, emit_alive(c->stream, (const uint2 *)c->uf_sd.p, n_seeds + 1, (uint32_t *)c->alive.p, fe.emit_grid, d_records, cap, mf + MF_LAKE_COUNT,
                            levels - 1));
    else
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
        // A group of two million records and more (2048^2 planes on) crosses the bus as u32 -- a colour and an area of a plane below
        // 2^31 pixels fit -- and is widened into the caller's records by the host's threads while later levels are computed
        // (ws_hostcopy.hip; pieces of at most n records: the narrowed words borrow the u64 label buffer, n x 8 bytes).
        // A piece that is too short for the chunked road must NOT go down it: labels_to_host_u64's other path widens into the very
        // buffer the narrowed words sit in (and may reallocate it).  Such pieces are copied as they are.
        const bool may_narrow = n < 0x80000000ull && c->out64.p && c->out64.cap >= n * sizeof(uint64_t);
        while (copied < end) {
          const size_t piece = std::min<size_t>(end - copied, n);
          if (may_narrow && 2 * piece >= ((size_t)1 << 22) && host_copy_in_chunks(c, 2 * piece)) {      // (from 2 M records a piece: below, starting the threads costs what they save -- 1024^2: 4.5 against 4.1 ms)
            HIP_TRY(c, narrow_words(c->copy_stream, (const uint64_t *)((const ws_lake *)c->lakes.p + copied), (uint32_t *)c->out64.p, 2 * piece));
            if ((rc = labels_to_host_u64(c, (const uint32_t *)c->out64.p, (uint64_t *)(lakes + copied), 2 * piece, c->copy_stream))) return rc;
          } else {
            HIP_TRY(c, hipMemcpyAsync(lakes + copied, (const ws_lake *)c->lakes.p + copied, piece * sizeof(ws_lake), hipMemcpyDeviceToHost, c->copy_stream));
          }
          copied += piece;
        }
      }
    }
    HIP_TRY(c, hipStreamSynchronize(c->copy_stream));
    *n_lakes = offsets[levels];
    for (uint32_t l = 0; l < levels; ++l) uncoloured[l] = n - bounds[l + 1];                   // index 0 of lib.rs:630's vector
  } else {
    HIP_TRY(c, hipMemcpyAsync(bounds.data(), mf + MF_OFF_PX, 2 * (NLEVELS + 1) * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  }
  if (out_labels && n) {
    if (merging && !host_copy_in_chunks(c, n)) {
      HIP_TRY(c, relabel_u64(c->stream, keys, seg, parent, d_out64, n, opt->max_water_level));
      HIP_TRY(c, hipMemcpyAsync(out_labels, d_out64, n * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    } else {
      // a large plane crosses the bus as u32 and is widened by host threads (ws_hostcopy.hip); the relabelled u32 plane of
      // the merging transform borrows the u64 buffer, which that path does not use
      if (merging) HIP_TRY(c, relabel_u32(c->stream, keys, seg, parent, (uint32_t *)d_out64, n, opt->max_water_level));
      if ((rc = labels_to_host_u64(c, merging ? (const uint32_t *)d_out64 : seg, out_labels, n))) return rc;
    }
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
  if (padded && h * w == 0 && (rc = empty_image_block(c, &src, &src_stride, h, w))) return rc;
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

int ws_lists_from_arrival_device(ws_ctx *c, int merging, const uint32_t *d_keys, const uint32_t *d_seg_labels, size_t h, size_t w,
                                 size_t n_seeds, const ws_options *opt, ws_lake *d_lakes, size_t cap, size_t *n_lakes, uint64_t *offsets,
                                 uint64_t *uncoloured) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c) return WS_ERR_BAD_ARG;
  if (!opt || !n_lakes || !offsets || !uncoloured || (!d_lakes && cap) || ((!d_keys || !d_seg_labels) && h * w))
    return fail(c, WS_ERR_BAD_ARG, "null pointer");
  if (pick_engine(opt) == WS_ENGINE_SWEEP) return fail(c, WS_ERR_UNSUPPORTED, "the sweep engine keeps no arrival stamps");
  if (int v = ws_options_validate(opt)) return fail(c, v, ws_strerror(v));
  if (h * w == 0) {      // no pixel: no lake at any level
    *n_lakes = 0;
    for (uint32_t l = 0; l <= opt->max_water_level; ++l) { offsets[l] = 0; uncoloured[l] = 0; }
    offsets[(size_t)opt->max_water_level + 1] = 0;
    return WS_OK;
  }
  DeviceLists dev{nullptr, nullptr, d_lakes};
  dev.d_keys = d_keys; dev.d_seg = d_seg_labels;
  return merge_host(c, merging != 0, nullptr, h, w, w, nullptr, n_seeds, opt, nullptr, nullptr, nullptr, nullptr, cap, n_lakes, offsets,
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
