/*
 * ws_oracle.h -- CPU restatement of the rustronomy-watershed hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it,
 * and only as the checker / the timed CPU baseline.  The shipped library
 * (rustronomy-watershed_amd/csrc) never links, loads or calls this code.
 *
 * Every function cites the reference lines it restates ("lib.rs:N" means
 * /root/reference/src/lib.rs line N, crate v0.4.1).  No reference source is
 * copied: the reference is Rust, this is a from-behaviour C restatement.
 *
 * Pinning: the reference holds only 7 inline unit tests for this path
 * (lib.rs:259-291, 308-311, 336-344, 369-377, 447-465, 544-587, 594-626).
 * Their 8x8 vectors are transcribed as data in tests/golden/reference_unit_vectors.json
 * and tests/test_oracle_golden.py checks this oracle against every one of them.
 * The reference has NO end-to-end golden outputs and cannot be built here (no
 * Rust toolchain), and its segmenting tie-break is random (lib.rs:249-253), so
 * whole-transform parity with the reference itself is "parity unpinned"; it is
 * defined instead as (a) bit-exactness with this deterministic restatement
 * (tie-break = col0, lib.rs:245 -- a legal outcome of lib.rs:251) and (b) the
 * reachable-sample check ws_or_check_reachable() below.
 */
#ifndef WS_ORACLE_H
#define WS_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* lib.rs:138-141 */
#define WS_OR_UNCOLOURED 0u
#define WS_OR_NORMAL_MAX 254u
#define WS_OR_ALWAYS_FILL 0u
#define WS_OR_NEVER_FILL 255u

/* tie-break used where a pixel touches lakes of different colours (lib.rs:244-254) */
#define WS_OR_TIE_FIRST 0  /* col0: first coloured neighbour in D,R,L,U order (lib.rs:245) */
#define WS_OR_TIE_RANDOM 1 /* uniform pick among coloured neighbours, with multiplicity (lib.rs:251-253) */

/* colour-map representative (lib.rs:467-542) */
#define WS_OR_MAP_FAITHFUL 0  /* region[0] exactly as the reference builds its regions (order dependent) */
#define WS_OR_MAP_CANONICAL 1 /* smallest colour of the merged class (union-find); same partition */

#define WS_OR_OK 0
#define WS_OR_ERR_SEED_OOB (-1) /* the reference panics (ndarray index) at lib.rs:1366 / 1676 */
#define WS_OR_ERR_ALLOC (-2)

typedef struct ws_or_stats {
  uint64_t scans;        /* calls of the flood step (find_flooded_px), incl. the empty ones */
  uint64_t max_rings;    /* largest number of non-empty rings within one level */
  uint64_t flooded;      /* pixels coloured by the flood (excludes seeds) */
  uint64_t conflicts;    /* flooded pixels that hit the tie-break branch (lib.rs:249-253) */
  uint64_t merge_pairs;  /* merging only: total unordered pairs found over all levels */
} ws_or_stats;

/* Per-level hook, the C shape of HookCtx (lib.rs:844-862).  `labels` is the
 * (padded, when edge correction is on) label plane after this level. */
typedef void (*ws_or_level_cb)(void *user, uint8_t water_level, uint8_t max_water_level,
                               const uint8_t *img, const uint64_t *labels, size_t h, size_t w);

/* Synthetic field of the bench/tests: v = mix64((seed << 40) + pixel_index) % 254,
 * i.e. iid uniform on [0,254) like Uniform::new(0,254) (README.md:60, tests/core_bench.rs:29). */
uint64_t ws_or_mix64(uint64_t x);
void ws_or_random_field(uint8_t *img, size_t h, size_t w, uint64_t seed);

/* lib.rs:196-257: one synchronous flood step.  Returns the number of emitted pixels;
 * out_rc holds (row, col) pairs, out_col the colour, out_conflict (nullable) 1 where the
 * tie-break branch was taken.  Capacity of each output: h*w entries. */
size_t ws_or_find_flooded_px(const uint8_t *img, const uint64_t *cols, size_t h, size_t w,
                             uint8_t lvl, int tie_mode, uint64_t *rng_state, uint64_t *out_rc,
                             uint64_t *out_col, uint8_t *out_conflict);

/* lib.rs:1638-1808: segmenting driver (transform_with_hook).  out_labels has
 * (h+2e)*(w+2e) entries, e = edge_correction?1:0.  arr_level/arr_ring (nullable, same
 * shape) receive each pixel's arrival time: level -1 for seeds, -2 for never coloured. */
int ws_or_segment(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc, size_t n_seeds,
                  uint8_t max_water_level, int edge_correction, int tie_mode, uint64_t rng_seed,
                  uint64_t *out_labels, int32_t *arr_level, uint32_t *arr_ring, ws_or_level_cb cb,
                  void *user, ws_or_stats *stats);

/* lib.rs:393-445: adjacent-label pair detection.  Returns the number of distinct
 * unordered pairs, written as (min,max) sorted ascending; cap = capacity in pairs. */
size_t ws_or_find_merge(const uint64_t *labels, size_t h, size_t w, uint64_t *out_pairs, size_t cap);

/* lib.rs:467-542: closure of pair mergers into the persistent colour map. */
int ws_or_make_colour_map(uint64_t *base_map, size_t map_len, const uint64_t *pairs, size_t n_pairs,
                          int map_mode);

/* lib.rs:589-592 */
void ws_or_recolour(uint64_t *labels, size_t n, const uint64_t *colour_map);

/* lib.rs:628-635: hist has n+1 entries (n = pixel count of the plane). */
void ws_or_find_lake_sizes(const uint64_t *labels, size_t n, uint64_t *hist);

/* lib.rs:1328-1522: merging driver (transform_with_hook). */
int ws_or_merge(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc, size_t n_seeds,
                uint8_t max_water_level, int edge_correction, int tie_mode, uint64_t rng_seed,
                int map_mode, uint64_t *out_labels, ws_or_level_cb cb, void *user,
                ws_or_stats *stats);

/* lib.rs:1524-1536: MergingWatershed::transform is a stub (zeros, interior = 123). */
void ws_or_merge_transform_stub(size_t h, size_t w, uint64_t *out_labels);

/* lib.rs:1178-1197: strict 8-neighbour local MAXIMA of the interior, row-major order.
 * Returns the count; writes at most cap (row, col) pairs. */
size_t ws_or_find_local_minima(const uint8_t *img, size_t h, size_t w, uint64_t *out_rc, size_t cap);

/* Reachable-sample check: is `cand` a possible output of the randomised
 * reference for these inputs?  0 = yes; otherwise 1 (coloured mask differs) or
 * 2 (a label is not the label of any earlier-arrived 4-neighbour) or 3 (seed label
 * wrong), and *bad_index (nullable) is the first offending pixel. */
int ws_or_check_reachable(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds, uint8_t max_water_level, int edge_correction,
                          const uint64_t *cand, size_t *bad_index);

/* Canonical relabelling for the merging transform: every label class gets the
 * smallest seed colour whose seed pixel lies in the class (seed colours as painted
 * by lib.rs:1360-1367).  In place.  Returns the number of classes. */
size_t ws_or_canonicalise(uint64_t *labels, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds);

/* lib.rs:1081-1173: pre_processor / pre_processor_with_max.  dtype: 0 f32, 1 f64, 2 i32, 3 u16, 4 i16,
 * 5 u8.  Returns 0, or -1 when max_value is outside 1..=254 (the reference asserts). */
int ws_or_pre_processor(const void *data, int dtype, size_t n, uint8_t max_value, uint8_t *out);

/* ---- second, independent restatement: the arrival-time form ------------------
 * T(p) = max((img[p],1), succ(min_q T(q))) over the 4 neighbours, seeds = -inf,
 * label(p) = label(first q in D,R,L,U with T(q) < T(p)).  Computed with a Dijkstra
 * order.  Derived from lib.rs:196-257 + 1689-1748; tests prove it equal to
 * ws_or_segment on every case the sweep oracle can run.  Used as the fast checker
 * for large fields.  arr_key (nullable): (level << 32 | ring), 0 for seeds,
 * UINT64_MAX for never coloured. */
int ws_or_segment_arrival(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds, uint8_t max_water_level, int edge_correction,
                          uint64_t *out_labels, uint64_t *arr_key);

/* Merging through arrival levels + union-find: per-level canonical partition.
 * For each level l in 0..=max the callback receives the canonical label plane
 * (equal to ws_or_merge(..., CANONICAL) followed by ws_or_canonicalise). */
int ws_or_merge_arrival(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                        size_t n_seeds, uint8_t max_water_level, int edge_correction,
                        uint64_t *out_labels, ws_or_level_cb cb, void *user);

#ifdef __cplusplus
}
#endif
#endif
