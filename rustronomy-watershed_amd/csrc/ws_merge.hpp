// ws_merge.hpp -- launch wrappers of the merging-transform kernels (ws_merge.hip).
//
// The reference's merging driver (lib.rs:1328-1522) floods exactly like the segmenting one
// and, after every level, merges every pair of touching lakes (find_merge lib.rs:393-445,
// make_colour_map 467-542, recolour 589-592).  Because a flooded pixel always takes the
// colour of an already coloured neighbour, the lakes after level l are the connected
// components of the pixels coloured by level l, and -- expressed over the SEGMENTING
// result -- the classes of a union-find over seed colours joined by every adjacent pixel
// pair (p, q) with different segmenting colours and max(level(p), level(q)) <= l, where
// at least one of p, q is interior (find_merge only looks at 3x3 window centres).
// The canonical lake id is the smallest colour of the class = the union-by-min root.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace wsk {

constexpr int NLEVELS = 256;

typedef unsigned long long u64c;

// parent[i] = i for i in 0..n, size[i] = 0
hipError_t uf_init(hipStream_t s, uint32_t *parent, uint32_t *size, size_t n);

// per-level histograms of arriving pixels and of label-crossing edges (256 bins each)
hipError_t level_hist(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                      u64c *hist_px, u64c *hist_edge);
// bucketed scatter: px_items[cursor_px[lvl]++] = colour, edge_items[cursor_edge[lvl]++] = (a,b)
hipError_t level_scatter(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                         u64c *cursor_px, u64c *cursor_edge, uint32_t *px_items, uint2 *edge_items);

// exclusive prefix sums of the histograms (NLEVELS + 1 entries each) and the scatter cursors, all on the device
hipError_t level_offsets(hipStream_t s, const u64c *hist_px, const u64c *hist_ed, u64c *off_px, u64c *off_ed, u64c *cur_px, u64c *cur_ed);
// the per-level kernels on a bucket whose bounds {first, end} sit in device memory (range = off + level): fixed grid,
// grid-stride loops -- nothing the host has to read before it launches, so the whole level loop can be captured
hipError_t union_edges_ranged(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned grid, uint32_t *parent, uint32_t *hooked,
                              uint32_t *hooked_count);
hipError_t fold_and_add_ranged(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items,
                               const u64c *range, unsigned grid, uint32_t *parent, uint32_t *size);

// lock-free union-by-min of n edges; every node that loses its root status is appended to hooked
hipError_t union_edges(hipStream_t s, const uint2 *edges, size_t n, uint32_t *parent, uint32_t *hooked,
                       uint32_t *hooked_count);
// areas of the nodes hooked in this level to their roots (hooked may be null: nothing was hooked) + arriving pixels counted
hipError_t fold_and_add(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items, size_t n,
                        uint32_t *parent, uint32_t *size);
// appends (colour, area) of every root with area > 0 behind the records of the earlier levels; counts past cap too
// level_counts: one record counter per level, zero on entry; the records of level l start at the sum of
// level_counts[0 .. l) (earlier launches) and level_counts[l] receives this level's count
// death (nullable): per colour, the level at which it stopped being a root (union_emit keeps it); then "lake of `level`"
// means death[c] > level instead of parent[c] == c
hipError_t emit_lakes(hipStream_t s, const uint32_t *parent, const uint32_t *size, size_t n_colours,
                      uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t level, const uint32_t *death = nullptr);
// the unions of `level` (bucket bounds in device memory) and the lake records of level - 1 in one launch; death: n_colours
// words, 0xFFFFFFFF at the start
hipError_t union_emit(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned union_grid, uint32_t *parent, uint32_t *hooked,
                      uint32_t *hooked_count, uint32_t *death, uint32_t level, const uint32_t *size, size_t n_colours, uint64_t *lakes,
                      size_t cap, u64c *level_counts);

// merging transform_to_list at size (ws_merge.hip, "records from the list of LIVE lakes"): sd[c] = (area, death level);
// alive: two lists of alive_list_words(n_colours) words; level L reads list (L + 1) & 1 (level 0: every colour) and writes list L & 1
hipError_t sd_init(hipStream_t s, uint2 *sd, size_t n);      // (0, 0xFFFFFFFF)
size_t alive_list_words(size_t n_colours);                   // words of ONE of the two live lists
hipError_t union_emit_alive(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned union_grid, uint32_t *parent, uint32_t *hooked,
                            uint32_t *hooked_count, uint2 *sd, uint32_t level, size_t n_colours, uint32_t *alive, unsigned emit_grid,
                            uint64_t *lakes, size_t cap, u64c *level_counts);
hipError_t emit_alive(hipStream_t s, const uint2 *sd, size_t n_colours, uint32_t *alive, unsigned emit_grid, uint64_t *lakes, size_t cap,
                      u64c *level_counts, uint32_t L);
hipError_t fold_and_add_sd(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items, const u64c *range,
                           unsigned grid, uint32_t *parent, uint2 *sd);

// merging across the row blocks of a tiled field: joins the touching colours of one block (seam pairs to its halo rows
// included; row0 = field row of the block's first local row, H = rows of the whole field), and the (colour, root) pairs
// of the block's boundary and halo rows (4 * w of them) that the ranks exchange
hipError_t block_union_pixels(hipStream_t s, const uint32_t *labels, int h, int w, int row0, int H, uint32_t *parent, int col0 = 0, int W = -1);      // (a tile: its first column in the field, the field's width)
hipError_t block_colour_roots2d(hipStream_t s, const uint32_t *labels, int h, int w, uint32_t *parent, uint2 *pairs, size_t n_pairs);      // rows and columns: 4 w + 4 h pairs, (0, 0) after them
hipError_t block_colour_roots(hipStream_t s, const uint32_t *labels, int h, int w, uint32_t *parent, uint2 *pairs);

// final-only path: union every crossing edge of the whole image in one launch
// final level only (coloured <=> label != 0); tile_min: union_image_tiles(h, w) words of scratch
size_t union_image_tiles(int h, int w);
hipError_t union_image(hipStream_t s, const uint32_t *labels, const uint32_t *seeds_rc, size_t n_seeds, int h, int w,
                       uint32_t *parent, uint32_t *tile_min, bool preclassified = false,      // preclassified: resolve_two_launch filled tile_min
                       uint32_t *tile_root_mark = nullptr);      // optional: n_seeds + 1 zeroed words (k_union_seeds' plain-store hooks)
// flattens the forest (n_colours entries, entry 0 = "uncoloured" points at itself) and gathers out[i] = root(labels[i])
hipError_t relabel_final_u32(hipStream_t s, const uint32_t *labels, uint32_t *parent, size_t n_colours, uint32_t *out, size_t n,
                             uint32_t *tile_min = nullptr, int h = 0, int w = 0);      // tile_min: union_image's classification of the h x w plane's tiles (one-lake tiles are filled)
// out[p] = coloured by `level` ? root(labels[p]) : 0
hipError_t relabel_u32(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *parent,
                       uint32_t *out, size_t n, uint32_t level);
hipError_t relabel_u64(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *parent,
                       uint64_t *out, size_t n, uint32_t level);

}  // namespace wsk
