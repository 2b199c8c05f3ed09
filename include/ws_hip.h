/*
 * ws_hip.h -- C ABI of the MI355X (gfx950) watershed engine.
 *
 * This is the drop-in boundary for ONE path of smups/rustronomy-watershed v0.4.1:
 * the segmenting / merging watershed transform and its seed finder.  The reference
 * has no FFI of its own (it is a single Rust file); the seam this ABI replaces is
 * the body of the private drivers behind the `Watershed<T>` trait.  Every entry
 * point cites the reference interface it stands in for ("lib.rs:N" =
 * src/lib.rs line N of the reference).  The Rust-side binding a maintainer would
 * add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types;
 *   - images are row-major u8 with a row stride in bytes (>= w);
 *   - seeds are (row, col) pairs, two uint64_t per seed (Rust `(usize, usize)` repacked);
 *   - host-side label planes are uint64_t (Rust `usize` on x86-64), row-major,
 *     shape (h + 2e) x (w + 2e) with e = 1 when edge correction is on (the
 *     reference's hook/transform outputs stay padded: lib.rs:1640-1666, 1812-1818);
 *   - device-side label planes are uint32_t (colours are 1..=n_seeds < 2^32);
 *   - every function returns WS_OK (0) or a negative ws_status; nothing aborts.
 *   - a ws_ctx is single-threaded; different contexts may be used from different
 *     host threads concurrently (the reference's structs are Send + Sync: lib.rs:65-67).
 *
 * Semantics fixed by the reference and reproduced here
 *   - find_local_minima returns strict 8-neighbour local MAXIMA of the interior,
 *     in row-major order (lib.rs:1178-1197);
 *   - colours are 1..=n_seeds in slice order, later duplicates overwrite (lib.rs:1670-1677);
 *   - only interior pixels are flooded (3x3 windows, lib.rs:220-222);
 *   - levels run 0..=max_water_level inclusive (lib.rs:1689);
 *   - where lakes of different colours meet, the reference picks a random neighbour
 *     colour (lib.rs:249-253); this engine always takes the first coloured neighbour
 *     in the reference's own order down,right,left,up (lib.rs:190, 245) -- a legal
 *     outcome of the reference, and the only tie rule offered (WS_TIE_FIRST_DRLU);
 *   - SegmentingWatershed::transform as written panics (lib.rs:1821 indexes the
 *     level-0 hook result); ws_segment returns the intended result, the labels after
 *     the last level (what transform_history(..).last() yields, lib.rs:1824-1835);
 *   - merged-lake ids DIFFER IN VALUE from the reference's: there a merged lake takes region[0] of
 *     make_colour_map's closure (lib.rs:508-541), which is a function of the order of the pair list
 *     (parallel unstable sort + dedup, lib.rs:440-443) -- deterministic for a given list, but not a
 *     property of the lake.  This engine returns the canonical id: the smallest seed colour whose
 *     seed pixel lies in the lake.  The PARTITION into lakes is the reference's (tested against the
 *     oracle's faithful closure under random tie-breaks); the id values are a documented deviation,
 *     "parity unpinned" by the reference (it has no test or fixture for them).
 */
#ifndef WS_HIP_H
#define WS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WS_ABI_VERSION 3 /* 2: ws_options grew to 8 bytes (seed_shift); 3: ws_group_* / ws_*_tiled* / ws_segment_batch* / ws_segment_minima* / ws_lists_from_arrival_device / ws_ctx_set_host_threads, WS_ERR_RCCL */

/* lib.rs:138-141 */
#define WS_UNCOLOURED 0u
#define WS_NORMAL_MAX 254u
#define WS_ALWAYS_FILL 0u
#define WS_NEVER_FILL 255u

typedef enum ws_status {
  WS_OK = 0,
  WS_ERR_BAD_ARG = -1,       /* null pointer, stride < w, ... */
  WS_ERR_MAX_TOO_HIGH = -2,  /* BuildErr::MaxToHigh, lib.rs:1026-1027 */
  WS_ERR_MAX_TOO_LOW = -3,   /* BuildErr::MaxToLow,  lib.rs:1028-1029 */
  WS_ERR_SEED_OOB = -4,      /* the reference panics: lib.rs:1366 / 1676 */
  WS_ERR_HIP = -5,           /* a HIP runtime call failed; see ws_last_error */
  WS_ERR_OOM = -6,           /* device or host allocation failed */
  WS_ERR_NO_DEVICE = -7,
  WS_ERR_CAPACITY = -8,      /* output buffer too small; *n_found still holds the true count */
  WS_ERR_RING_OVERFLOW = -9, /* > 2^24-1 flood rings inside one level (needs > 16.7 M pixel corridor) */
  WS_ERR_TOO_LARGE = -10,    /* plane has >= 2^32 pixels or seeds */
  WS_ERR_UNSUPPORTED = -11,
  WS_ERR_RCCL = -12          /* an RCCL call failed, or librccl.so could not be loaded; see ws_group_last_error */
} ws_status;

typedef enum ws_engine {
  WS_ENGINE_AUTO = 0,
  /* all water levels fused: every pixel carries its (level, ring) arrival stamp and the
   * flood is relaxed tile-by-tile in LDS; bit-identical to the sweep */
  WS_ENGINE_FUSED = 1,
  /* literal level sweep: one synchronous 4-neighbour flood step per launch, repeated
   * until unchanged, for every level (lib.rs:1689-1748 one-to-one) */
  WS_ENGINE_SWEEP = 2
} ws_engine;

#define WS_TIE_FIRST_DRLU 0

/* The runtime options of TransformBuilder (lib.rs:908-923) as plain data. */
typedef struct ws_options {
  uint8_t max_water_level; /* lib.rs:950; valid 1..=254 */
  uint8_t edge_correction; /* lib.rs:958; 0/1.  The (h+2) x (w+2) plane of lib.rs:1640-1666 is only ever the shape of the
                              label plane: the ring of zeros around the image is virtual, no padded image copy is made. */
  uint8_t engine;          /* ws_engine */
  uint8_t tie_rule;        /* WS_TIE_FIRST_DRLU */
  uint8_t seed_shift;      /* only with edge_correction.  0 (default): seeds index the PADDED plane with the caller's
                              coordinates, i.e. land one pixel up-left of where they were found -- what the reference does
                              (lib.rs:1364-1366, 1675-1677: `output[*seed_idx]` on the padded array).  1: every seed is moved by
                              (+1, +1) onto its own pixel -- what edge correction evidently intends; not reference behaviour. */
  uint8_t reserved[3];     /* must be zero */
} ws_options;

typedef struct ws_ctx ws_ctx;

/* Counters of the last transform run on a context (the reference's `debug` PerfReport,
 * lib.rs:640-696, restated for this engine). */
typedef struct ws_stats {
  uint32_t relax_passes;      /* fused engine: global relaxation passes launched */
  uint32_t resolve_passes;    /* fused engine: global label-resolve passes launched */
  uint32_t sweep_steps;       /* sweep engine: flood steps launched */
  uint32_t merge_levels;      /* merging: levels with at least one union */
  uint64_t tiles_run_relax;   /* tiles that actually did work, summed over passes, in units of 8192 pixels (a seam band of 256 x 8 is a quarter) */
  uint64_t tiles_run_resolve;
  float ms_relax;             /* HIP-event time per kernel class; filled only when */
  float ms_resolve;           /* profiling is enabled with ws_ctx_set_profiling     */
  float ms_sweep;
  float ms_other;
  float ms_total;             /* device time of the whole call between first and last launch */
  uint32_t launches_relax;
  uint32_t launches_resolve;
  uint32_t launches_sweep;
  uint32_t relax_tile_iterations; /* k_relax2: in-tile sweeps summed over tiles and passes */
  uint32_t graph_launches;    /* 1: the call's seed tables, first passes and gated resolve ran as one hipGraph launch
                                 (a transform repeating the previous one's buffers, sizes and seed count, on a
                                 non-null stream); occupies what was tail padding: the struct size is unchanged */
} ws_stats;

/* HookCtx (lib.rs:844-862) as a C callback, invoked once per water level, in order.
 * `labels` is a host plane valid only during the call.  The seeds slice of HookCtx is
 * the caller's own seed list, so it is not passed back. */
typedef void (*ws_level_cb)(void *user, uint8_t water_level, uint8_t max_water_level,
                            const uint8_t *image, const uint64_t *labels, size_t h, size_t w);

/* One lake of one level for transform_to_list (lib.rs:628-635, 1551-1561): the dense
 * Vec<usize> of length h*w+1 per level is returned sparsely. */
typedef struct ws_lake {
  uint64_t colour;
  uint64_t area;
} ws_lake;

/* ---- context ------------------------------------------------------------------------- */

int ws_abi_version(void);
const char *ws_strerror(int status);

/* Creates a context on HIP device `device` with its own stream. */
int ws_ctx_create(int device, ws_ctx **out);
/* Same, but all work is enqueued on the caller's hipStream_t (e.g. PyTorch's current
 * stream), so the caller's events bracket the kernels. */
int ws_ctx_create_on_stream(int device, void *hip_stream, ws_ctx **out);
void ws_ctx_destroy(ws_ctx *ctx);
const char *ws_last_error(const ws_ctx *ctx);
int ws_ctx_set_profiling(ws_ctx *ctx, int enabled);
int ws_ctx_get_stats(const ws_ctx *ctx, ws_stats *out);
int ws_ctx_synchronize(ws_ctx *ctx);
/* ws_segment_batch_device stacks at most this many pixels into one transform (the stack needs 9 bytes per pixel of
 * context workspace); larger batches run as several stacks.  0 restores the default, 2^31 - 1, which is also the cap. */
int ws_ctx_set_batch_pixel_limit(ws_ctx *ctx, size_t max_px);
/* The segmenting transform of a plane with at least this many pixels repairs the seams of its first relaxation pass with
 * bands and strips instead of a second pass over every tile (same labels either way).  0 restores the
 * default, 2^24: smaller planes are bound by launch gaps and gain nothing.  (Tests lower it to cover the path on small planes.) */
int ws_ctx_set_seam_repair_min_pixels(ws_ctx *ctx, size_t min_px);
/* Long-range floods (smooth maps: a flood crosses thousands of pixels and the relaxation needs a hundred passes and more): with
 * mode 1 or 2 the first pass of the late, same-grid regime is ONE persistent launch in which workgroups pull 128 x 64 tiles from
 * a device-side queue and a tile that changes something its neighbour must see queues that neighbour at once (stamps handed
 * over write-through / past L1 at agent scope); the pass after it looks at every tile again, so the labels are the same in
 * every mode.  1: first come, first served (a ring).  2: in flood order -- 31 buckets by the level of the smallest stamp that
 * waits at a tile's borders, the lowest non-empty bucket first, and a run that announces a tile of its own bucket takes it
 * itself; it runs on 128 x 128 tiles and starts at pass 3.  0: the ordinary passes.  3 (the default): mode 2 when the seeds
 * are sparse -- fewer than one per two tiles, so that floods cross several tiles each -- the passes otherwise.  Measured on
 * 8192^2 smooth maps of correlation length 4 / 16 / 64 / 256 px (683 k / 8.6 k / 35 / 1 seeds): 3.0 / 5.8 / 6.7 / 3.5 ms with the
 * passes, 3.0 / 6.4 / 6.4 / 4.1 with mode 1, 4.2 / 5.8 / 3.9 / 3.2 with mode 2, 3.0 / 5.8 / 3.9 / 3.2 with the default
 * (profiles/r3_v1_persistent_ab.txt).  4: the passes on their EARLY schedule (one grid from pass 3,
 * scans from pass 2), what the default picks for between one seed per two tiles and ~30 per tile (5.8 -> 5.3 ms at 16 px).
 * Transforms that converge in a few passes (random fields) never reach those passes.  WS_ERR_BAD_ARG for any other mode. */
int ws_ctx_set_persistent_pass(ws_ctx *ctx, int mode);
/* Host threads the host-buffer entry points may start for the length of a call (default 4, at most 64, never more than the
 * machine has): from 2^21 pixels on a u64 label plane (ws_segment, ws_segment_minima, ws_merge ...) crosses the bus as the device's
 * u32 plane in chunks (a quarter of the plane, 16 MiB at most) and these threads widen each chunk into the caller's memory while the next ones are in flight --
 * half the bytes over PCIe (8192^2: ws_segment_minima 11.5 -> 7.2 ms).  0: no threads, the plane is widened on the device and
 * copied whole, as before. */
int ws_ctx_set_host_threads(ws_ctx *ctx, int n_threads);
/* The merging transform_to_list of a seed list with at least this many entries writes every level's lake records from the
 * list of the lakes alive at the level before, instead of looking at every colour at every level (same records, the order
 * inside a level differs).  0 restores the default, 2^20: fewer colours are bound by launch latency and gain nothing.
 * (Tests lower it to cover the form on small planes.) */
int ws_ctx_set_live_list_min_colours(ws_ctx *ctx, size_t min_colours);

/* TransformBuilder::build_segmenting / build_merging validation (lib.rs:999-1004, 1026-1030). */
int ws_options_default(ws_options *out);          /* lib.rs:936-946: max 254, no edge correction */
int ws_options_validate(const ws_options *opt);

/* ---- host-buffer entry points (what the Rust shim binds) ---------------------------- */

/* WatershedUtils::find_local_minima (lib.rs:1178-1197).  Writes up to `cap` (row, col)
 * pairs; *n_found receives the total.  WS_ERR_CAPACITY when cap is too small. */
int ws_find_local_minima(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride,
                         uint64_t *out_rc, size_t cap, size_t *n_found);

/* Watershed::transform for SegmentingWatershed (lib.rs:1810-1822, intended semantics).
 * out_labels is the reference's Array2<usize> plane.  From 2^21 pixels on the labels cross the bus as the device's u32 plane, in
 * chunks, and host threads of the library (ws_ctx_set_host_threads: four; 0 = none, one 8-byte copy) widen them into
 * out_labels while the next chunks are in flight: 8192^2 9.2 ms instead of 13.7.  The threads live for
 * the call only. */
int ws_segment(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride,
               const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt,
               uint64_t *out_labels);

/* The same with the labels as the device holds them, uint32_t (colours are 1..=n_seeds < 2^32): one plain copy, no host
 * threads, half the bytes in the caller's memory (8192^2: 8.7 ms; the transform itself is 0.56 ms of it, the rest is the link at
 * its 53 GB/s).  For callers that can hold u32 labels; not what transform() returns. */
int ws_segment_u32(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride,
                   const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt,
                   uint32_t *out_labels);

/* The README's call pair as ONE call (lib.rs:73-86: `let mins = ws.find_local_minima(img); ws.transform(img, &mins)`):
 * the seeds are find_local_minima(img) (lib.rs:1178-1197, in its row-major order: colour i + 1 for its i-th entry) and the labels
 * are transform's for that list.  The list never has to exist: with the fused engine, no edge correction and w % 32 == 0 the
 * seed tables come out of the minima kernels themselves; otherwise the two calls run one after the other.  seeds_rc (cap
 * pairs, may be NULL with cap 0) receives the list when the caller wants it; *n_seeds its length.  WS_ERR_CAPACITY when
 * cap > 0 is too small -- the labels are complete then, the list is cut.  8192^2 host form: 117 MB of seed pairs less each
 * way over PCIe than ws_find_local_minima + ws_segment. */
int ws_segment_minima(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride, const ws_options *opt,
                      uint64_t *out_labels, uint64_t *seeds_rc, size_t cap, size_t *n_seeds);
int ws_segment_minima_u32(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride, const ws_options *opt,
                          uint32_t *out_labels, uint64_t *seeds_rc, size_t cap, size_t *n_seeds);
/* ... on device-resident buffers: d_labels h x w (padded with edge correction) uint32_t, d_seeds_rc cap (row, col) uint32_t
 * pairs or NULL. */
int ws_segment_minima_device(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride, const ws_options *opt,
                             uint32_t *d_labels, uint32_t *d_seeds_rc, size_t cap, size_t *n_seeds);

/* Watershed::transform_with_hook for SegmentingWatershed (lib.rs:1638-1808): cb is called
 * after every level 0..=max with the label plane of that level; transform_history
 * (lib.rs:1824-1835) is this with a copying callback.  out_labels may be NULL. */
int ws_segment_with_hook(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride,
                         const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt,
                         ws_level_cb cb, void *user, uint64_t *out_labels);

/* Watershed::transform_with_hook for MergingWatershed (lib.rs:1328-1522).  Labels passed
 * to cb / written to out_labels carry canonical representatives (see top of file). */
int ws_merge_with_hook(ws_ctx *ctx, const uint8_t *img, size_t h, size_t w, size_t row_stride,
                       const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt,
                       ws_level_cb cb, void *user, uint64_t *out_labels);

/* Watershed::transform_to_list (lib.rs:1551-1561 merging, 1837-1847 segmenting), sparse:
 * for level l the lakes with area > 0 are lakes[offsets[l] .. offsets[l+1]), every lake
 * once, as runs of increasing colours in no particular order of the runs (the reference's
 * vector is indexed by colour, so a caller scatters them: hist[colour] = area); the
 * uncoloured count (index 0 of the reference's vector) is uncoloured[l].
 * offsets has max_water_level+2 entries, uncoloured max_water_level+1.
 * *n_lakes receives the total number of records; WS_ERR_CAPACITY if cap is too small. */
int ws_transform_to_list(ws_ctx *ctx, int merging, const uint8_t *img, size_t h, size_t w,
                         size_t row_stride, const uint64_t *seeds_rc, size_t n_seeds,
                         const ws_options *opt, ws_lake *lakes, size_t cap, size_t *n_lakes,
                         uint64_t *offsets, uint64_t *uncoloured);

/* MergingWatershed::transform is a stub in the reference (lib.rs:1524-1536): zeros with the
 * interior set to 123, seeds ignored.  Reproduced for drop-in completeness. */
int ws_merge_transform_stub(size_t h, size_t w, uint64_t *out_labels);

/* A cube of n_slices independent h x w slices in HOST memory -- tests/integration.rs:267,356: the reference calls
 * find_local_minima + transform for one slice of a CGPS cube after the other.  Slice k starts at cube + k * slice_stride; its
 * seeds are seeds_rc[2 * seed_offsets[k] .. 2 * seed_offsets[k + 1]) (seed_offsets: n_slices + 1 entries), or, with seeds_rc ==
 * NULL, its own find_local_minima (lib.rs:1178-1197; n_seeds[k], nullable, receives how many); its labels go to out_labels +
 * k * (padded plane).  Equivalent to n_slices calls of ws_segment (ws_segment_minima) and bit-identical to them; the slices take
 * turns on four internal contexts with a stream each and a host thread each for the length of the call, so that one slice's
 * upload, another's transform and a third's label copy overlap on the two directions of the link: 16 x 4096^2 in 23 ms against
 * 34 ms for the loop (1.44 ms a slice; its labels alone take the link 1.17).  The context's statistics (ws_ctx_get_stats) and last
 * arrival stamps are not those of any slice afterwards.  On an error the lowest failing slice is reported in *failed_slice (nullable), its message through ws_last_error. */
int ws_segment_batch(ws_ctx *ctx, const uint8_t *cube, size_t n_slices, size_t h, size_t w, size_t row_stride,
                     size_t slice_stride, const uint64_t *seeds_rc, const size_t *seed_offsets, const ws_options *opt,
                     uint64_t *out_labels, size_t *n_seeds, size_t *failed_slice);

/* ---- device-resident entry points (inputs and outputs stay in HBM) ------------------- */

/* d_img: device u8 plane; d_seeds_rc: device (row, col) pairs as uint32_t[2];
 * d_labels: device uint32_t plane of the (padded, if edge correction) shape.
 * Any seed list is accepted (duplicates: the later entry wins, lib.rs:1672-1677).  A list in strictly
 * increasing row-major order -- what ws_find_local_minima(_device) returns -- takes a faster path; the
 * engine checks the order on the device, nothing has to be declared. */
int ws_find_local_minima_device(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w,
                                size_t row_stride, uint32_t *d_out_rc, size_t cap,
                                size_t *n_found);
int ws_segment_device(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                      const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt,
                      uint32_t *d_labels);
/* ws_segment_device in two halves, for pipelines: _begin queues the transform and returns, _end waits for it and
 * returns its status.  Between the two the context belongs to that transform (no other call on it) and the caller's
 * buffers must stay as they are.  Only a transform that repeats the previous call's arguments on this context (same
 * buffers, sizes and seed count: its launches are replayed as one graph) is actually left in flight; any other runs
 * whole inside _begin.  Contexts taking turns keep the GPU's queue from running dry between transforms (created on one
 * stream: 8192^2 0.55 -> 0.53 ms per transform), and on streams of their own their transforms overlap on the GPU, one
 * filling the CUs another leaves idle (four contexts: 0.45 ms per transform; a single transform still takes 0.55). */
int ws_segment_device_begin(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                            const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt,
                            uint32_t *d_labels);
int ws_segment_device_end(ws_ctx *ctx);
/* A stack of independent slices (BASELINE config C4: a cube cut into 2-D slices, as the reference's own
 * integration tests do, tests/integration.rs:267,356): slice k starts at d_cube + k * slice_stride, its
 * seeds are d_seeds_rc[2 * seed_offsets[k] .. 2 * seed_offsets[k+1]) (seed_offsets: n_slices + 1 entries, on
 * the HOST), its labels go to d_labels + k * (padded plane).  Equivalent to n_slices calls of
 * ws_segment_device; stops at the first slice that fails and reports its index in *failed_slice.
 * Contiguous slices (slice_stride == h * row_stride, row_stride == w unless edge correction copies them anyway) with
 * strictly increasing seed lists, w' % 4 == 0 and h' * w' % 128 == 0 (h', w': the padded plane) run as ONE transform
 * over the stacked slices -- the border rows of a slice never flood, so they wall the slices off from each other --
 * instead of n_slices transforms: 16 x 1024^2 in 0.28 ms instead of 2.2 ms.  Results are identical either way.
 * After a stacked batch ws_last_arrival_device reports "unsupported" (the stamps are those of the stack). */
int ws_segment_batch_device(ws_ctx *ctx, const uint8_t *d_cube, size_t n_slices, size_t h, size_t w,
                            size_t row_stride, size_t slice_stride, const uint32_t *d_seeds_rc,
                            const size_t *seed_offsets, const ws_options *opt, uint32_t *d_labels,
                            size_t *failed_slice);
/* Merging transform, final canonical labels only. */
int ws_merge_device(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                    const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt,
                    uint32_t *d_labels);
/* ... in two halves, as ws_segment_device_begin / _end (same rules). */
int ws_merge_device_begin(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                          const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt,
                          uint32_t *d_labels);
int ws_merge_device_end(ws_ctx *ctx);
/* Watershed::transform_to_list with everything in HBM: the image, the u32 seed pairs and the lake RECORDS (d_lakes: cap
 * records in the caller's device buffer; at 1024^2 they are 155 MB that the host form spends most of its time copying).
 * Only the per-level offsets (max_water_level + 2) and uncoloured counts (max_water_level + 1) go to the host arrays.
 * Same record layout, order and WS_ERR_CAPACITY protocol as ws_transform_to_list. */
int ws_transform_to_list_device(ws_ctx *ctx, int merging, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                                const uint32_t *d_seeds_rc, size_t n_seeds, const ws_options *opt, ws_lake *d_lakes,
                                size_t cap, size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured);

/* The same lists from a segmenting transform that has already run -- here or on other devices: d_keys = its arrival stamps
 * (ws_last_arrival_device / ws_copy_last_arrival_device), d_seg_labels = its labels, both h x w u32 planes as they stand (a
 * padded plane is passed padded; opt->edge_correction is ignored), n_seeds = the number of colours.  Everything the per-level
 * paths read is in those two planes: no image, no seed list, no second flood.  Records, offsets, uncoloured and
 * WS_ERR_CAPACITY as ws_transform_to_list_device. */
int ws_lists_from_arrival_device(ws_ctx *ctx, int merging, const uint32_t *d_keys, const uint32_t *d_seg_labels, size_t h, size_t w,
                                 size_t n_seeds, const ws_options *opt, ws_lake *d_lakes, size_t cap, size_t *n_lakes,
                                 uint64_t *offsets, uint64_t *uncoloured);
/* Arrival stamps of the last ws_segment_device / ws_merge_device call on this context:
 * (level << 24 | ring), 0 for seeds, 0xFF000000 for never coloured.  Device pointer owned
 * by the context, valid until the next call; shape as the label plane. */
int ws_last_arrival_device(ws_ctx *ctx, const uint32_t **d_keys, size_t *h, size_t *w);
/* Copies those stamps into the caller's device buffer of n_elems >= h*w words (stream ordered). */
int ws_copy_last_arrival_device(ws_ctx *ctx, uint32_t *d_dst, size_t n_elems);
/* transform_history on the device, one level at a time (lib.rs:1824-1835 without the 255 host copies): the SEGMENTING
 * label plane as the reference's hook sees it after `water_level` (lib.rs:1796-1804) -- a pixel carries its final colour
 * once its arrival level is <= water_level, 0 before -- from d_labels (what the last ws_segment_device on this context
 * wrote) and the context's arrival stamps; d_out is a plane of the same shape (may equal d_labels).  Stream ordered. */
int ws_level_snapshot_device(ws_ctx *ctx, const uint32_t *d_labels, uint8_t water_level, uint32_t *d_out);

/* ---- input preparation (first "next" row) ---------------------------------------
 *
 * WatershedUtils::pre_processor / pre_processor_with_max::<MAX> (lib.rs:1081-1173): any numeric
 * array (any dimension, passed flattened) -> u8 in [0, MAX].  Quirks reproduced as coded: min and
 * max folds are seeded with 0; values that are not `is_normal` in f64 -- NaN, -inf, subnormals AND
 * exact 0 -- become NEVER_FILL (255); +inf becomes ALWAYS_FILL (0); the rest is
 * trunc((x - min) / (max - min) * MAX) in f64.  max_value must be in 1..=254 (the reference asserts). */
typedef enum ws_dtype { WS_F32 = 0, WS_F64 = 1, WS_I32 = 2, WS_U16 = 3, WS_I16 = 4, WS_U8 = 5 } ws_dtype;
int ws_pre_processor(ws_ctx *ctx, const void *data, int dtype, size_t n_elems, uint8_t max_value, uint8_t *out);
int ws_pre_processor_device(ws_ctx *ctx, const void *d_data, int dtype, size_t n_elems, uint8_t max_value,
                            uint8_t *d_out);

/* ---- one field tiled over several GPUs: row blocks with halo rows ----------------------
 *
 * A rank holds its rows of the global field plus one extra (halo) row on every side that has a
 * neighbour.  The local plane's first and last rows are then either the global border or a halo,
 * which the flood never writes (lib.rs:220-222), so a block is relaxed exactly like a whole
 * image.  The caller alternates: ws_block_relax on every rank -> exchange halo rows of d_keys
 * (RCCL send/recv) -> all-reduce of `changed`, until no rank changed; then the same loop with
 * ws_block_resolve and d_labels.  Seeds carry GLOBAL colours (index in the caller's slice + 1,
 * lib.rs:1670-1672) and LOCAL coordinates; every rank paints the seeds that fall on any of its
 * local rows, halo rows included.  rustronomy-watershed_amd/distributed.py drives this. */
int ws_block_init(ws_ctx *ctx, size_t h, size_t w, const uint32_t *d_seeds_rc, const uint32_t *d_colours,
                  size_t n_seeds, uint32_t *d_keys, uint32_t *d_labels);
int ws_block_relax(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                   uint8_t max_water_level, uint32_t *d_keys, int *changed);
int ws_block_resolve(ws_ctx *ctx, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w, int *changed);
/* A tile of a field cut in both directions: the labels of the whole plane from painted labels, in two launches.  Seeds hold
 * their colours; the plane's border ring holds what is known of the neighbours' pixels so far (0: nothing yet) and is a set
 * of roots like the seeds.  The caller swaps the ring and repeats until nobody receives anything new (ws_segment_tiled2d*
 * do). */
int ws_block_resolve_ring(ws_ctx *ctx, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w);

/* The same block in its fast form, for seed lists in strictly increasing row-major order (what ws_find_local_minima
 * returns): a rank's seeds -- those on any of its local rows, halo rows included -- are then entries [g0, g0 + n) of the
 * caller's list, in local coordinates, and first_colour = g0 + 1.  Needs w % 4 == 0 and h * w < 2^31; anything else
 * (WS_ERR_UNSUPPORTED) goes through ws_block_init / _relax / _resolve above.  Sequence on every rank (distributed.py):
 *   ws_block_begin;  repeat { swap halo rows of d_keys with the neighbours; stop when no rank received a row that
 *   differs from what it held; ws_block_relax_halo };  ws_block_resolve_local;  ws_block_export_boundary;  all-gather the
 *   2 * w words of every rank into one table;  ws_block_import_boundary.
 * Stamps need that iteration (a flood may cross a seam more than once); labels do not: after the local resolve every
 * label is a colour or a reference to a neighbour's boundary-row pixel, the boundary rows of all ranks form a closed
 * table (entry (rank r, side s, column x) at (2 r + s) * w + x; s = 0 first own row, 1 last own row; a reference is
 * 0x80000000 | entry index), and every rank resolves the gathered table for itself. */
int ws_block_begin(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride, uint8_t max_water_level,
                   const uint32_t *d_seeds_rc, size_t n_seeds, uint32_t first_colour, uint32_t *d_keys);
int ws_block_relax_halo(ws_ctx *ctx, const uint8_t *d_img, size_t h, size_t w, size_t row_stride, uint8_t max_water_level,
                        int halo_top, int halo_bottom, uint32_t *d_keys);
int ws_block_resolve_local(ws_ctx *ctx, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w, int halo_top,
                           int halo_bottom);
int ws_block_export_boundary(ws_ctx *ctx, const uint32_t *d_labels, size_t h, size_t w, int halo_top, int halo_bottom,
                             size_t rank, uint32_t *d_rows /* 2 * w words */);
int ws_block_import_boundary(ws_ctx *ctx, const uint32_t *d_table /* world * 2 * w words */, size_t world, size_t rank,
                             uint32_t *d_labels, size_t h, size_t w, int halo_top, int halo_bottom);

/* The MERGING transform's final canonical labels of a tiled field (lib.rs:1328-1522 over row blocks), after the tiled
 * segmenting transform has left d_labels (halo rows included) final.  d_parent: n_colours_total + 1 words (colour 0 =
 * uncoloured), a union-find over ALL seed colours of the field, owned by the caller.  row0 = field row of the block's
 * first local row.  Sequence: ws_block_merge_local; ws_block_merge_export (4 * w (colour, root) pairs: the block's two
 * boundary rows and its halo rows); all-gather of the pairs; ws_block_merge_import; ws_block_merge_relabel.  One
 * exchange, no rounds: a lake that spans several blocks is a chain of local pieces linked at boundary colours. */
int ws_block_merge_local(ws_ctx *ctx, const uint32_t *d_labels, size_t h, size_t w, size_t row0, size_t field_rows,
                         size_t n_colours_total, uint32_t *d_parent);
int ws_block_merge_export(ws_ctx *ctx, const uint32_t *d_labels, size_t h, size_t w, uint32_t *d_parent, uint32_t *d_pairs);
int ws_block_merge_import(ws_ctx *ctx, const uint32_t *d_pairs, size_t n_pairs, uint32_t *d_parent);
int ws_block_merge_relabel(ws_ctx *ctx, const uint32_t *d_labels, size_t n, uint32_t *d_parent, size_t n_colours_total,
                           uint32_t *d_out);

/* ---- several GPUs driven inside the library --------------------------------------------------------
 *
 * The reference's drivers are one address space and one rayon pool (lib.rs:1689-1748); a drop-in that wants the GPUs of a
 * node behind the same transform() call cannot ask its caller to bring torch.distributed or MPI.  A ws_group is a set of
 * RANKS, each with a context of its own on one device, plus the three exchange steps the tiled transform needs -- swap the
 * halo rows with the neighbour ranks, max-reduce one word, all-gather a small table -- in one of two implementations:
 *
 *   local  all ranks live in THIS process (one host thread each while a call runs); rank r sits on devices[r].  The
 *          exchange steps are stream-ordered copies between the ranks' buffers (peer-to-peer between devices).  A device
 *          may appear more than once: ranks then share it -- the protocol rehearsed on a one-GPU box, and what the tests run.
 *   rccl   THIS process is ONE rank of `world` (one process per GPU: torchrun / mpirun).  Halo rows travel as grouped
 *          ncclSend / ncclRecv pairs, the flag word through ncclAllReduce(max), tables through ncclAllGather, all on the
 *          rank's own stream, over xGMI.  librccl.so is loaded when the first such group is created.  Every rank passes the
 *          same 128-byte id, made by ONE rank with ws_group_rccl_unique_id and handed to the others by whatever the job
 *          already uses (MPI_Bcast, a file, torch.distributed.broadcast).
 *
 * Either way the transform is the single-domain one bit for bit (the arrival-stamp fixpoint is unique, DESIGN.md section
 * 2 and 6): row blocks with one halo row per neighbour; stamps iterate (relax to local convergence, swap halo rows, stop
 * when no rank received a row that differs from the one it held -- the difference is found on the device and reduced
 * there, one host read per round); labels need one all-gather of the ranks' boundary rows (2 * w words per rank). */
typedef struct ws_group ws_group;
#define WS_RCCL_ID_BYTES 128

int ws_group_create_local(int n_ranks, const int *devices /* n_ranks entries; NULL: every rank on device 0 */, ws_group **out);
int ws_group_rccl_unique_id(void *id /* WS_RCCL_ID_BYTES */);
int ws_group_create_rccl(int device, int rank, int world, const void *id /* WS_RCCL_ID_BYTES */, ws_group **out);
void ws_group_destroy(ws_group *g);
/* world: ranks of the whole group; n_local: ranks driven by this process (local: all; rccl: 1); first_local: the first of them */
int ws_group_info(const ws_group *g, int *world, int *n_local, int *first_local);
const char *ws_group_last_error(const ws_group *g);
/* Runs every exchange step of the group once on small buffers and checks the results (WS_OK, or WS_ERR_RCCL / WS_ERR_HIP):
 * a deployment check that the transport works before a field is committed to it.  Collective: every rank calls it. */
int ws_group_selftest(ws_group *g);

/* Rows of a field of `h` rows that rank `rank` of `world` OWNS: [*r0, *r1); its local plane is rows [*lo, *hi) = the same
 * plus one halo row on every side that has a neighbour.  WS_ERR_BAD_ARG when h < world (every rank must own a row). */
int ws_tile_rows(size_t h, int rank, int world, size_t *r0, size_t *r1, size_t *lo, size_t *hi);

/* One field tiled over the ranks of a group, host buffers in and out -- what a caller of transform(ArrayView2<u8>, &seeds)
 * (lib.rs:1810) has.  Every rank of the group calls it with the SAME arguments (local: one call drives all ranks); rank r
 * uploads its rows of the image, takes its part of the seed list (any list: one that is not strictly increasing, or a
 * width that is not a multiple of 4, takes the general form -- painted seeds, iterative label rounds -- on all ranks
 * together), and writes the rows it owns of out_labels -- for a local group that is all of them.  merging != 0: the
 * MERGING transform's final canonical labels (lib.rs:1328-1522 after the last level).  Edge correction pads the field
 * first, as the reference does (lib.rs:1640-1666): out_labels is (h + 2) x (w + 2) then.  *exchange_rounds (nullable):
 * collective steps of the call. */
int ws_segment_tiled(ws_group *g, const uint8_t *img, size_t h, size_t w, size_t row_stride, const uint64_t *seeds_rc,
                     size_t n_seeds, const ws_options *opt, int merging, uint64_t *out_labels, uint32_t *exchange_rounds);

/* The same with everything resident in HBM: one descriptor per LOCAL rank (ws_group_info), in rank order. */
typedef struct ws_tile_block {
  const uint8_t *d_img;        /* the rank's local plane: rows [lo, hi) of the field (ws_tile_rows), row stride w, on the rank's device */
  const uint32_t *d_seeds_rc;  /* the seeds that fall on ANY local row, halo rows included, in LOCAL coordinates (row - lo, col) */
  const uint32_t *d_colours;   /* their colours (index in the caller's list + 1, lib.rs:1670-1672); NULL: first_colour, first_colour + 1, ...
                                  -- a contiguous range of a strictly increasing list, which is what the fast form needs */
  size_t n_seeds;
  uint32_t first_colour;
  uint32_t reserved;           /* must be zero */
  uint32_t *d_labels;          /* out: the local plane's labels, (hi - lo) x w u32; rows 0 / last are halo rows where the rank has neighbours */
} ws_tile_block;
int ws_segment_tiled_device(ws_group *g, size_t field_h, size_t w, size_t n_seeds_total, const ws_tile_block *blocks,
                            const ws_options *opt /* edge_correction must be 0: pad the field first */, int merging,
                            uint32_t *exchange_rounds);

/* transform_to_list (lib.rs:1551-1561 merging, 1837-1847 segmenting) of a field in row blocks (blocks as for
 * ws_segment_tiled_device, whose segmenting transform runs first and leaves every block's labels in d_labels): the ranks then
 * send the arrival stamps and labels of their owned rows to rank 0 (8 B per pixel, one message per rank and plane), and rank
 * 0's context writes the lake records of all levels from the whole plane (ws_lists_from_arrival_device).  d_lakes (cap
 * records, on rank 0's device), n_lakes, offsets (max_water_level + 2) and uncoloured (max_water_level + 1) are filled in the
 * process that holds rank 0; elsewhere *n_lakes = 0 and the arrays are left alone.  A lake spans blocks and its area is a sum
 * over them: the lists are global, so they are made where the whole plane is -- 28 B per pixel of one device's 288 GB. */
int ws_transform_to_list_tiled_device(ws_group *g, size_t field_h, size_t w, size_t n_seeds_total, const ws_tile_block *blocks,
                                      const ws_options *opt, int merging, ws_lake *d_lakes, size_t cap, size_t *n_lakes,
                                      uint64_t *offsets, uint64_t *uncoloured, uint32_t *exchange_rounds);
/* ... and with host buffers, as ws_segment_tiled takes them (edge correction by padding: the lists are those of the padded
 * plane, as ws_transform_to_list's): every rank uploads its rows, the records come back from rank 0's device into `lakes`
 * (cap records; WS_ERR_CAPACITY with the count in *n_lakes when that is too few).  What a Rust caller of transform_to_list gets
 * from several GPUs. */
int ws_transform_to_list_tiled(ws_group *g, int merging, const uint8_t *img, size_t h, size_t w, size_t row_stride,
                               const uint64_t *seeds_rc, size_t n_seeds, const ws_options *opt, ws_lake *lakes, size_t cap,
                               size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured, uint32_t *exchange_rounds);

/* The field cut in BOTH directions (BASELINE config 5's "2-D tiles"): py x px tiles, rank = ty * px + tx, py * px = the
 * group's ranks.  ws_tile_grid: rows[4] = {r0, r1, lo, hi} and cols[4] = {c0, c1, clo, chi} of a rank's tile -- owned
 * [r0, r1) x [c0, c1), held [lo, hi) x [clo, chi) with a halo row / column on every side that has a neighbour (ws_tile_rows
 * in each direction).  One descriptor per LOCAL rank; seeds carry their colours (index in the caller's list + 1,
 * lib.rs:1670-1672: a tile's seeds are no contiguous range of the list).  The exchange is halo rows AND columns, 2 (w + h)
 * words a tile and round; the block steps are the general form's (painted seeds, relaxation rounds, label rounds: the
 * seed-table / boundary-table shortcuts of the row-block form are not built for tiles).  merging != 0: the MERGING transform's
 * final canonical labels (one more gather: (colour, root) pairs of every tile's outermost rows and columns). */
typedef struct ws_tile_block2d {
  const uint8_t *d_img;        /* the tile's plane [lo, hi) x [clo, chi) of the field, on the rank's device */
  size_t img_stride;           /* >= chi - clo (a view into the whole field works) */
  const uint32_t *d_seeds_rc;  /* seeds on ANY pixel of the plane, halo ring included, LOCAL coordinates (row - lo, col - clo) */
  const uint32_t *d_colours;   /* their colours */
  size_t n_seeds;
  uint32_t *d_labels;          /* out: (hi - lo) x (chi - clo) u32 */
} ws_tile_block2d;
int ws_tile_grid(size_t h, size_t w, int rank, int py, int px, size_t *rows /* 4 */, size_t *cols /* 4 */);
int ws_segment_tiled2d_device(ws_group *g, size_t field_h, size_t field_w, int py, int px, size_t n_seeds_total,
                              const ws_tile_block2d *blocks, const ws_options *opt, int merging, uint32_t *exchange_rounds);
/* transform_to_list of a field in py x px tiles: ws_transform_to_list_tiled_device with the owned rectangles gathered on rank 0. */
int ws_transform_to_list_tiled2d_device(ws_group *g, size_t field_h, size_t field_w, int py, int px, size_t n_seeds_total,
                                        const ws_tile_block2d *blocks, const ws_options *opt, int merging, ws_lake *d_lakes,
                                        size_t cap, size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured,
                                        uint32_t *exchange_rounds);
/* ... with host buffers in and out, as ws_segment_tiled: every rank of the group calls it with the SAME arguments, uploads its
 * tile, takes its seeds (any list), writes the rectangle it owns of out_labels.  Edge correction pads the field first. */
int ws_segment_tiled2d(ws_group *g, const uint8_t *img, size_t h, size_t w, size_t row_stride, const uint64_t *seeds_rc,
                       size_t n_seeds, const ws_options *opt, int py, int px, int merging, uint64_t *out_labels,
                       uint32_t *exchange_rounds);

/* BASELINE config C4 over a group: a batch of independent slices, slice i on rank i % world, a rank's slices as ONE stacked
 * transform (ws_segment_batch_device) -- no exchange step at all.  One descriptor per LOCAL rank: the rank's own slices,
 * contiguous in its HBM.  Local groups run their ranks side by side (a host thread each). */
typedef struct ws_batch_part {
  const uint8_t *d_cube;         /* n_slices slices of h x w, contiguous (slice stride h * w), on the rank's device */
  const uint32_t *d_seeds_rc;    /* all of its slices' (row, col) pairs, concatenated */
  const size_t *seed_offsets;    /* n_slices + 1 entries, on the HOST */
  size_t n_slices;
  uint32_t *d_labels;            /* n_slices planes */
} ws_batch_part;
int ws_segment_batch_group(ws_group *g, size_t h, size_t w, const ws_batch_part *parts, const ws_options *opt,
                           size_t *failed_rank, size_t *failed_slice);

/* The same batch from HOST memory (ws_segment_batch above) over the ranks of a group: rank r takes the slices
 * [n_slices r / world, n_slices (r + 1) / world) of the cube and pipelines them on its own device -- every device on its own link
 * to the host, the config-4 shape for a caller whose cube lives in host memory.  Arguments as ws_segment_batch (seed_offsets and
 * n_seeds index the WHOLE cube); a process drives its local ranks only: in an RCCL group, its one rank's block of slices, read
 * from and written to that process's pointers.  No exchange step.  *failed_slice: the lowest failing slice of the local ranks. */
int ws_segment_batch_host(ws_group *g, const uint8_t *cube, size_t n_slices, size_t h, size_t w, size_t row_stride,
                          size_t slice_stride, const uint64_t *seeds_rc, const size_t *seed_offsets, const ws_options *opt,
                          uint64_t *out_labels, size_t *n_seeds, size_t *failed_slice);

/* Bench/test synthetic field: v = mix64((seed << 40) + index) % 254. */
int ws_random_field_device(ws_ctx *ctx, uint8_t *d_img, size_t h, size_t w, size_t row_stride,
                           uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
