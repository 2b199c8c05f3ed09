// ws_block_api.hip -- row blocks of one larger field, one C call per step: the caller (distributed.py over
// torch.distributed, or any MPI-style host) brings the collectives.  ws_tiled.hip drives the same steps inside the library.
#include "ws_ctx.hpp"

using namespace wsapi;

extern "C" {

// ---- row blocks of one larger field (multi-GPU tiling) -----------------------------------------
//
// A rank owns a block of rows of the global field and holds it with one extra row on every side
// that has a neighbour rank.  First/last local rows are therefore either the global border or a
// halo copy, i.e. exactly the rows the flood never writes (lib.rs:220-222), so the single-GPU
// kernels run unchanged on the local plane; the caller exchanges halo rows between calls.

int ws_block_init(ws_ctx *c, size_t h, size_t w, const uint32_t *d_seeds_rc, const uint32_t *d_colours, size_t n_seeds,
                  uint32_t *d_keys, uint32_t *d_labels) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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

// A TILE of a field cut in both directions (ws_segment_tiled2d*): labels of the whole plane in two launches from painted
// labels -- seeds hold their colours, the plane's border ring holds what is known of the neighbours' pixels so far (0:
// nothing yet) and is a set of roots like the seeds.  Repeated after every swap of the ring until no rank receives anything
// new: a round carries labels across one tile boundary, against one HOP per launch of the iterative form
// (8192^2 in 2 x 2 tiles: 32 launches of k_resolve per rank and step, 1.2 of its 2.4 ms).
int ws_block_resolve_ring(ws_ctx *c, const uint32_t *d_keys, uint32_t *d_labels, size_t h, size_t w) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || (h * w && (!d_keys || !d_labels))) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if (h > 0x7FFFFFF0ull || w > 0x7FFFFFF0ull || h * w >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "plane has >= 2^31 pixels");
  if (h * w == 0) return WS_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  int rc;
  if ((rc = ensure(c, c->refs, resolve_ref_capacity((int)h, (int)w) * sizeof(uint32_t)))) return rc;
  uint32_t *flags = (uint32_t *)c->flags.p;
  HIP_TRY(c, hipMemsetAsync(flags + FLAG_OVERFLOW, 0, sizeof(uint32_t), c->stream));
  c->misc_clean = false;
  HIP_TRY(c, resolve_two_launch(c->stream, d_keys, d_labels, (int)h, (int)w, (uint32_t *)c->refs.p, c->debug_max_iters, nullptr, nullptr,
                                nullptr, nullptr, 0, flags + FLAG_OVERFLOW, nullptr, 4));
  HIP_TRY(c, hipMemcpyAsync(&c->pinned[FLAG_OVERFLOW], flags + FLAG_OVERFLOW, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->pinned[FLAG_OVERFLOW]) return fail(c, WS_ERR_RING_OVERFLOW, "more than 2^24-1 flood rings inside one level");
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_labels || !d_rows) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  if ((halo_top && rank == 0) || h < (size_t)(1 + (halo_top ? 1 : 0) + (halo_bottom ? 1 : 0))) return fail(c, WS_ERR_BAD_ARG, "halo flags do not fit the block");
  if ((rank + 2) * 2 * w >= 0x80000000ull) return fail(c, WS_ERR_TOO_LARGE, "boundary table index needs more than 31 bits");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_export_boundary(c->stream, d_labels, (int)h, (int)w, (halo_top ? 1 : 0) | (halo_bottom ? 2 : 0), (uint32_t)rank, d_rows));
  return WS_OK;
}

int ws_block_import_boundary(ws_ctx *c, const uint32_t *d_table, size_t world, size_t rank, uint32_t *d_labels, size_t h, size_t w,
                             int halo_top, int halo_bottom) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
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
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_labels || !d_parent || !d_pairs || h == 0) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, block_colour_roots(c->stream, d_labels, (int)h, (int)w, d_parent, (uint2 *)d_pairs));
  return WS_OK;
}

int ws_block_merge_import(ws_ctx *c, const uint32_t *d_pairs, size_t n_pairs, uint32_t *d_parent) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_parent || (n_pairs && !d_pairs)) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  // (0, 0) pairs -- uncoloured boundary pixels -- join colour 0 with itself: nothing happens
  HIP_TRY(c, union_edges(c->stream, (const uint2 *)d_pairs, n_pairs, d_parent, nullptr, nullptr));
  return WS_OK;
}

int ws_block_merge_relabel(ws_ctx *c, const uint32_t *d_labels, size_t n, uint32_t *d_parent, size_t n_colours_total, uint32_t *d_out) {
  if (int busy_rc = refuse_if_in_flight(c)) return busy_rc;
  if (!c || !d_parent || (n && (!d_labels || !d_out))) return fail(c, WS_ERR_BAD_ARG, "bad argument");
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, relabel_final_u32(c->stream, d_labels, d_parent, n_colours_total + 1, d_out, n));
  return WS_OK;
}

}  // extern "C"
