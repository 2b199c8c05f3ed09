// ws_common.hpp -- shared definitions of the gfx950 watershed kernels.
//
// Arrival stamps.  The reference floods level by level and, inside a level, ring by
// ring (lib.rs:1689-1748); a pixel coloured in ring r of level l gets the stamp
//     key = (l << 24) | r          (r >= 1),
// seeds carry key 0 (coloured before level 0, lib.rs:1675-1677) and pixels that are
// never coloured carry KEY_INF.  In this encoding the reference's flood step
// (lib.rs:224-231: flooded, uncoloured, a coloured 4-neighbour) becomes the fixpoint
//     key(p) = max(base(p), 1 + min over the 4 neighbours q of key(q)),
//     base(p) = (img[p] << 24) | 1 for interior pixels with img[p] <= max level,
// and the colour is the colour of the first neighbour in down,right,left,up order
// (lib.rs:190) whose key is smaller (lib.rs:237-248 with the col0 tie rule).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace wsk {

// Debug and A/B knobs (WS_DEBUG_MAXIT, WS_NO_GRAPH, WS_RELAX_P0_ROUNDS ...) exist only in a -DWS_TUNING build, which
// tools/ makes for its experiments.  The shipped library never reads its environment: an inherited variable must not
// be able to change what a drop-in transform returns.
#ifdef WS_TUNING
inline const char *tuning_env(const char *name) { return getenv(name); }
#else
inline const char *tuning_env(const char *) { return nullptr; }
#endif

constexpr uint32_t KEY_INF = 0xFF000000u;   // level 255 can never open (NEVER_FILL, lib.rs:141)
constexpr uint32_t RING_MASK = 0x00FFFFFFu;

constexpr int TS = 64;        // tile side of the label-resolve kernels
constexpr int LP = TS + 2;    // LDS tile side including the 1-px halo
constexpr int STRIP = 16;     // rows per thread there: 256 threads = 64 columns x 4 strips
constexpr int NTHREADS = 256;

// Per-pass convergence words.  Workgroups never share an atomic: a tile that changed stores 1
// (plain, idempotent) into the stripe blockIdx % NSTRIPE of the pass's slot; stripes sit on their
// own 64-byte lines.  The host reads one slot (NSTRIPE lines) per pass.
constexpr int COUNTER_RING = 64;                  // slots reused cyclically, one per pass (two groups of passes in flight: ws_ctx.hpp)
constexpr int NSTRIPE = 64;
constexpr int STRIPE_STRIDE = 16;                 // words between stripes (64 bytes)
constexpr int FLAG_SLOT = NSTRIPE * STRIPE_STRIDE;   // words per slot

// The hardware deals consecutive workgroups round-robin to the 8 XCDs, each with its own L2.  With the
// identity mapping horizontally adjacent tiles land on different XCDs and every halo column (one 4-byte
// pixel per row, a whole cache line fetched for it) misses: k_resolve_local fetched 2.35x its algorithmic
// bytes.  This gives each XCD one contiguous span of tile indices instead (tile rows, in row-major
// order), so a tile's neighbours run on the same L2 at about the same time.  Bijective on [0, n).
#ifdef __HIPCC__
// Inclusive prefix sum over the 64 lanes of a wave without LDS: four DPP row shifts inside the rows of 16 lanes (`old` = 0:
// a lane with nothing before it adds nothing), then the rows' totals by DPP row broadcasts (GFX9: row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3).  __shfl_up is a ds_bpermute: six dependent LDS round trips for the same sum.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
#define WS_DPP_ADD(ctrl, rows) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rows, 0xF, false)
  WS_DPP_ADD(0x111, 0xF);      // row_shr:1
  WS_DPP_ADD(0x112, 0xF);      // row_shr:2
  WS_DPP_ADD(0x114, 0xF);      // row_shr:4
  WS_DPP_ADD(0x118, 0xF);      // row_shr:8
  WS_DPP_ADD(0x142, 0xA);      // row_bcast:15
  WS_DPP_ADD(0x143, 0xC);      // row_bcast:31
#undef WS_DPP_ADD
  return v;
}

__device__ __forceinline__ uint32_t xcd_span_index(uint32_t b, uint32_t n) {
  constexpr uint32_t XCDS = 8;
  const uint32_t per = n / XCDS;
  return b < per * XCDS ? (b % XCDS) * per + b / XCDS : b;
}

// Edge correction (lib.rs:1640-1666) floods the image inside a ring of zeros.  The ring is never materialised: its
// pixels are the border pixels of the (h + 2) x (w + 2) plane, which no flood step ever writes (3x3 windows,
// lib.rs:220-222) -- their base is KEY_INF whatever the image says -- so only interior pixels' bytes are ever looked at,
// and logical pixel (y, x) of the plane is image pixel (y - 1, x - 1).  Border pixels read a clamped address.
// SH: rows per slice of a stacked batch (== H for one image); the image rows of slice k start at k * (SH - 2).
__device__ __forceinline__ size_t padded_img_index(int gy, int gx, int W, int SH, size_t stride) {
  const int slice = gy / SH, ry = gy - slice * SH;
  const int iy = max(min(max(ry, 1), SH - 2) - 1, 0), ix = max(min(max(gx, 1), W - 2) - 1, 0);
  return ((size_t)slice * (size_t)max(SH - 2, 0) + iy) * stride + ix;
}
#endif

struct PassFlags {
  uint32_t *edge_changed;   // [COUNTER_RING][FLAG_SLOT]: a tile edge changed in pass (p % COUNTER_RING)
  uint32_t *any_change;     // [FLAG_SLOT]: any pixel changed (row-block API)
  uint32_t *overflow;       // 1 word: ring field carried into the level field
  uint32_t *stats;          // nullptr unless profiling: [2][FLAG_SLOT] striped tile / sweep counters
};

// --- launch wrappers (ws_kernels.hip) ---------------------------------------------------
hipError_t fill_u32(hipStream_t s, uint32_t *p, size_t n, uint32_t v);
// host_a / host_b: DEVICE pointers to pinned host memory (hipHostGetDevicePointer)
hipError_t words_to_host(hipStream_t s, const uint32_t *a, uint32_t na, uint32_t *host_a, const uint32_t *b, uint32_t nb, uint32_t *host_b);
hipError_t zero3(hipStream_t s, uint32_t *a, size_t na, uint32_t *b, size_t nb, uint32_t *c, size_t nc);   // one launch
hipError_t random_field(hipStream_t s, uint8_t *img, size_t stride, int h, int w, uint64_t seed);
hipError_t scatter_seeds(hipStream_t s, const uint32_t *seeds_rc, const uint32_t *colours, size_t n, int ph, int pw,
                         uint32_t *labels, uint32_t *keys, uint32_t *err_flag);
hipError_t paint_labels(hipStream_t s, const uint32_t *seeds_rc, size_t n, int ph, int pw, uint32_t *labels,
                        uint32_t *err_flag, uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b);
hipError_t seed_tables(hipStream_t s, const uint32_t *seeds_rc, size_t n, int ph, int pw, uint32_t *mask, uint32_t *word_base,
                       uint32_t *err_flag, uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b,
                       const uint32_t *slice_first = nullptr, size_t slice_px = 0,      // stack of slices: colours restart in every slice
                       uint32_t colour_bias = 0);                                       // row block of a larger field: colour of list entry 0, minus 1
// seeds of a stack of slices -> seeds of the stacked plane (row + slice * slice_h); slice_first: n_slices + 1 list offsets
hipError_t stack_seeds(hipStream_t s, const uint32_t *seeds_rc, size_t n, const uint32_t *slice_first, size_t n_slices,
                       int slice_h, int pw, uint32_t *stacked_rc, uint32_t shift = 0);
hipError_t narrow_seeds(hipStream_t s, const uint64_t *src, size_t n, size_t ph, size_t pw, uint32_t *dst, uint32_t shift = 0);   // u64 pairs -> u32 pairs (+shift), out of plane -> ~0
hipError_t shift_seeds(hipStream_t s, const uint32_t *src, size_t n, uint32_t shift, uint32_t *dst);          // (r, c) -> (r + shift, c + shift)
hipError_t widen_labels(hipStream_t s, const uint32_t *src, uint64_t *dst, size_t n);
hipError_t snapshot_level(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint64_t *dst,
                          size_t n, uint32_t level);
hipError_t snapshot_level_u32(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *dst, size_t n, uint32_t level);

// relaxation (ws_relax.hip): 4x4 register patches, 256 x 32 tiles, row/column sweeps
size_t relax_tiles(int h, int w);
hipError_t relax_pass(hipStream_t s, const uint8_t *img, size_t img_stride, uint32_t *keys, int h, int w,
                      uint32_t max_level, uint32_t pass, uint32_t *stamps, PassFlags pf, uint32_t max_iters,
                      const uint32_t *seed_labels = nullptr,    // non-null: pass 0 derives the stamps from this label plane
                      bool seed_bits = false,                   // ... which is one bit per pixel (seed_tables) instead
                      int slice_h = 0,                          // > 0: the plane is a stack of independent slices of this many rows
                      bool carry_checked_later = false,         // the caller's resolve_two_launch(.., carry_flag) looks for ring carries
                      bool padded = false,                      // edge correction: img is the caller's (h-2) x (w-2) image, the ring of zeros is virtual
                      uint32_t *tile_list = nullptr,            // relax_list_words(h, w) words: late passes run from a compacted list of tiles
                      size_t seam_min_px = (size_t)1 << 24,     // planes from this many pixels on repair pass 0's seams with bands and strips (ws_relax.hip)
                      int persistent_pass = 0);                 // the late passes of a long-range flood as one persistent launch with a tile queue: 1 first come (from pass 7), 2 in flood order (from pass 3); the caller resolves "auto"
size_t relax_list_words(int h, int w);
bool relax_uses_seam_repair(int h, int w, bool seed_bits, int slice_h, bool padded, size_t seam_min_px);      // pass 1 of such a transform is two launches

// label resolve, iterative form (row blocks of a tiled field, planes >= 2^31 pixels): 64x64 tiles
size_t resolve_tiles(int h, int w);
hipError_t resolve_pass(hipStream_t s, const uint32_t *keys, uint32_t *labels, int h, int w,
                        uint32_t pass, uint32_t *stamps, PassFlags pf);
// label resolve without a launch loop (pointer jumping in LDS + reference chase); needs h*w < 2^31
// ref_scratch: resolve_ref_capacity(h, w) words (per-wave work lists of the pixels whose chain leaves their tile)
size_t resolve_ref_capacity(int h, int w);
hipError_t resolve_two_launch(hipStream_t s, const uint32_t *keys, uint32_t *labels, int h, int w, uint32_t *ref_scratch,
                              uint32_t max_rounds = 0xFFFFFFFFu,
                              const uint32_t *seed_mask = nullptr, const uint32_t *word_base = nullptr,    // seed_tables() form
                              uint32_t *tile_min = nullptr,       // merging: per 64x64 tile, one lake? + a colour of it (ws_merge.hpp)
                              const uint32_t *gate = nullptr,     // speculative launch: a pass's convergence slot; both kernels leave if it is set
                              int slice_h = 0,                    // > 0: stack of independent slices (see relax_pass)
                              uint32_t *carry_flag = nullptr,     // set when a stamp carried out of its ring field (relax_pass with fresh_keys leaves the test to this kernel)
                              const uint32_t *seed_err = nullptr,    // seed_tables()' three error words: both kernels leave when the tables are invalid
                              int halo_flags = 0);                   // row block: bit 0 / 1 = the first / last row is a neighbour's (chains stop there)
hipError_t resolve_chase_again(hipStream_t s, uint32_t *labels, int h, int w, uint32_t *ref_scratch);

// row blocks of one field tiled over ranks (ws_block.hip)
hipError_t block_flag_border_tiles(hipStream_t s, uint32_t *stamps, int h, int w, uint32_t pass, int halo_flags);
hipError_t block_export_boundary(hipStream_t s, const uint32_t *labels, int h, int w, int halo_flags, uint32_t rank, uint32_t *rows);
hipError_t block_import_boundary(hipStream_t s, const uint32_t *table, uint32_t world, uint32_t rank, uint32_t *resolved,
                                 uint32_t *labels, int h, int w, int halo_flags);
hipError_t block_iota(hipStream_t s, uint32_t *p, size_t n, uint32_t first);      // p[i] = first + i
hipError_t block_rows_differ(hipStream_t s, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *flag);      // raises *flag, never clears it
hipError_t block_pack_cols(hipStream_t s, const uint32_t *plane, size_t h, size_t w, size_t xl, size_t xr, uint32_t *out);      // out[0 .. h) = column xl, out[h .. 2h) = column xr
hipError_t block_unpack_cols(hipStream_t s, uint32_t *plane, size_t h, size_t w, const uint32_t *left, const uint32_t *right);      // columns 0 / w - 1 (null: left alone)

hipError_t flood_step(hipStream_t s, const uint8_t *img, size_t img_stride, const uint32_t *lin,
                      uint32_t *lout, int h, int w, uint32_t level, uint32_t *counter, bool padded = false);

// local maxima: count per 1024-px row segment, scan, write
size_t minima_segments(int h, int w);       // count words (one per row and one per strip of rows)
size_t minima_mask_bytes(int h, int w);     // nibble plane written by minima_count, read by minima_write
hipError_t minima_count(hipStream_t s, const uint8_t *img, size_t stride, int h, int w, uint32_t *counts, uint8_t *nibbles);
hipError_t minima_write(hipStream_t s, const uint8_t *nibbles, int h, int w, const uint32_t *counts, uint32_t *total, uint32_t *out_rc, size_t cap,      // *total: the list's length
                        uint32_t *mask = nullptr, uint32_t *word_base = nullptr,      // w % 32 == 0: the seed tables of this list (seed_tables' layout)
                        uint32_t *zero_a = nullptr, size_t n_zero_a = 0, uint32_t *zero_b = nullptr, size_t n_zero_b = 0);
hipError_t widen_pairs(hipStream_t s, const uint32_t *src, uint64_t *dst, size_t n_values);
hipError_t narrow_words(hipStream_t s, const uint64_t *src, uint32_t *dst, size_t n);      // 64-bit words known to fit 32 bits

inline int tiles_of(int n) { return (n + TS - 1) / TS; }

}  // namespace wsk
