/*
 * ws_oracle.c -- CPU restatement of the rustronomy-watershed hot path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY -- see ws_oracle.h.  "lib.rs:N" = /root/reference/src/lib.rs:N.
 *
 * Layout conventions: images are row-major u8, label planes row-major u64 (the
 * reference's `usize` on x86-64), seeds are (row, col) pairs of u64.
 */
#include "ws_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ PRNG -- */

uint64_t ws_or_mix64(uint64_t x) {
  /* splitmix64 output function */
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void ws_or_random_field(uint8_t *img, size_t h, size_t w, uint64_t seed) {
  const uint64_t base = seed << 40;
  for (size_t i = 0; i < h * w; ++i) img[i] = (uint8_t)(ws_or_mix64(base + (uint64_t)i) % 254u);
}

static uint64_t rng_next(uint64_t *state) {
  *state += 0x9E3779B97F4A7C15ull;
  return ws_or_mix64(*state);
}

/* ------------------------------------------------- neighbour order (a3) -- */

/* lib.rs:188-194: neighbours_4con yields (r+1,c), (r,c+1), (r,c-1), (r-1,c):
 * down, right, left, up.  This order fixes the tie-break. */
static const int N4_DR[4] = {+1, 0, 0, -1};
static const int N4_DC[4] = {0, +1, -1, 0};

/* lib.rs:170-185: neighbours_8con offsets (order irrelevant: used under `all`). */
static const int N8_DR[8] = {+1, +1, +1, 0, 0, -1, -1, -1};
static const int N8_DC[8] = {0, +1, -1, +1, -1, 0, +1, -1};

/* ------------------------------------------------------ flood step (a1) -- */

size_t ws_or_find_flooded_px(const uint8_t *img, const uint64_t *cols, size_t h, size_t w,
                             uint8_t lvl, int tie_mode, uint64_t *rng_state, uint64_t *out_rc,
                             uint64_t *out_col, uint8_t *out_conflict) {
  /* lib.rs:220-222: 3x3 windows => only interior centres; none when h<3 or w<3 */
  if (h < 3 || w < 3) return 0;
  size_t n = 0;
  for (size_t r = 1; r + 1 < h; ++r) {
    for (size_t c = 1; c + 1 < w; ++c) {
      const size_t p = r * w + c;
      if (img[p] > lvl) continue;                  /* lib.rs:224 flooded?   */
      if (cols[p] != WS_OR_UNCOLOURED) continue;   /* lib.rs:226 uncoloured? */
      uint64_t nb[4];
      int k = 0;
      for (int d = 0; d < 4; ++d) {                /* lib.rs:237-242 */
        const uint64_t q = cols[(r + N4_DR[d]) * w + (c + N4_DC[d])];
        if (q != WS_OR_UNCOLOURED) nb[k++] = q;
      }
      if (k == 0) continue;                        /* lib.rs:228-231 */
      int all_same = 1;
      for (int i = 1; i < k; ++i) all_same &= (nb[i] == nb[0]);
      uint64_t pick = nb[0];                       /* lib.rs:245-248 */
      if (!all_same && tie_mode == WS_OR_TIE_RANDOM) {
        pick = nb[rng_next(rng_state) % (uint64_t)k]; /* lib.rs:251-253 */
      }
      out_rc[2 * n] = r;
      out_rc[2 * n + 1] = c;
      out_col[n] = pick;
      if (out_conflict) out_conflict[n] = (uint8_t)!all_same;
      ++n;
    }
  }
  return n;
}

/* ------------------------------------------------ shared driver set-up -- */

typedef struct {
  size_t ph, pw;   /* plane shape (padded when edge correction is on) */
  uint8_t *pimg;   /* owned padded image or NULL */
  const uint8_t *img;
} plane_t;

/* lib.rs:1330-1356 / 1640-1666: optional 1-px zero padding of the input */
static int plane_setup(plane_t *pl, const uint8_t *img, size_t h, size_t w, int edge) {
  pl->pimg = NULL;
  if (!edge) {
    pl->ph = h; pl->pw = w; pl->img = img;
    return WS_OR_OK;
  }
  pl->ph = h + 2; pl->pw = w + 2;
  pl->pimg = (uint8_t *)calloc(pl->ph * pl->pw, 1);
  if (!pl->pimg) return WS_OR_ERR_ALLOC;
  for (size_t r = 0; r < h; ++r) memcpy(pl->pimg + (r + 1) * pl->pw + 1, img + r * w, w);
  pl->img = pl->pimg;
  return WS_OR_OK;
}

/* lib.rs:1360-1367 / 1670-1677: colours 1..=S in slice order; later duplicates
 * overwrite; seeds index the (padded) plane with UNSHIFTED coordinates; an
 * out-of-bounds seed panics in the reference. */
static int paint_seeds(uint64_t *labels, size_t ph, size_t pw, const uint64_t *seeds_rc, size_t n) {
  for (size_t i = 0; i < n; ++i) {
    const uint64_t r = seeds_rc[2 * i], c = seeds_rc[2 * i + 1];
    if (r >= ph || c >= pw) return WS_OR_ERR_SEED_OOB;
    labels[r * pw + c] = (uint64_t)i + 1;
  }
  return WS_OR_OK;
}

/* one level's colouring loop: lib.rs:1394-1438 / 1704-1748 */
static int colouring_loop(const plane_t *pl, uint64_t *labels, uint8_t lvl, int tie_mode,
                          uint64_t *rng, uint64_t *buf_rc, uint64_t *buf_col, uint8_t *buf_cf,
                          int32_t *arr_level, uint32_t *arr_ring, ws_or_stats *st) {
  uint64_t rings = 0;
  for (;;) {
    const size_t n = ws_or_find_flooded_px(pl->img, labels, pl->ph, pl->pw, lvl, tie_mode, rng,
                                           buf_rc, buf_col, buf_cf);
    if (st) st->scans++;
    if (n == 0) break;                                      /* lib.rs:1733-1735 */
    ++rings;
    for (size_t i = 0; i < n; ++i) {                        /* lib.rs:1741-1743 */
      const size_t p = buf_rc[2 * i] * pl->pw + buf_rc[2 * i + 1];
      labels[p] = buf_col[i];
      if (arr_level) arr_level[p] = (int32_t)lvl;
      if (arr_ring) arr_ring[p] = (uint32_t)rings;
      if (st) st->conflicts += buf_cf[i];
    }
    if (st) st->flooded += n;
  }
  if (st && rings > st->max_rings) st->max_rings = rings;
  return WS_OR_OK;
}

/* --------------------------------------------------- segmenting (a3) -- */

int ws_or_segment(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc, size_t n_seeds,
                  uint8_t max_water_level, int edge_correction, int tie_mode, uint64_t rng_seed,
                  uint64_t *out_labels, int32_t *arr_level, uint32_t *arr_ring, ws_or_level_cb cb,
                  void *user, ws_or_stats *stats) {
  plane_t pl;
  int rc = plane_setup(&pl, img, h, w, edge_correction);
  if (rc) return rc;
  const size_t n = pl.ph * pl.pw;
  if (stats) memset(stats, 0, sizeof *stats);
  memset(out_labels, 0, n * sizeof(uint64_t));              /* lib.rs:1647 */
  if (arr_level) for (size_t i = 0; i < n; ++i) arr_level[i] = -2;
  if (arr_ring) memset(arr_ring, 0, n * sizeof(uint32_t));
  rc = paint_seeds(out_labels, pl.ph, pl.pw, seeds_rc, n_seeds);
  if (rc) { free(pl.pimg); return rc; }
  if (arr_level) for (size_t i = 0; i < n; ++i) if (out_labels[i]) arr_level[i] = -1;

  uint64_t *buf_rc = (uint64_t *)malloc((n ? n : 1) * 2 * sizeof(uint64_t));
  uint64_t *buf_col = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint8_t *buf_cf = (uint8_t *)malloc(n ? n : 1);
  if (!buf_rc || !buf_col || !buf_cf) { free(buf_rc); free(buf_col); free(buf_cf); free(pl.pimg); return WS_OR_ERR_ALLOC; }
  uint64_t rng = rng_seed;

  for (unsigned lvl = 0; lvl <= max_water_level; ++lvl) {   /* lib.rs:1689: inclusive */
    colouring_loop(&pl, out_labels, (uint8_t)lvl, tie_mode, &rng, buf_rc, buf_col, buf_cf,
                   arr_level, arr_ring, stats);
    if (cb) cb(user, (uint8_t)lvl, max_water_level, pl.img, out_labels, pl.ph, pl.pw); /* lib.rs:1796-1804 */
  }
  free(buf_rc); free(buf_col); free(buf_cf); free(pl.pimg);
  return WS_OR_OK;
}

/* ------------------------------------------------------ find_merge (a6) -- */

static int cmp_pair(const void *a, const void *b) {
  const uint64_t *x = (const uint64_t *)a, *y = (const uint64_t *)b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  if (x[1] != y[1]) return x[1] < y[1] ? -1 : 1;
  return 0;
}

size_t ws_or_find_merge(const uint64_t *labels, size_t h, size_t w, uint64_t *out_pairs, size_t cap) {
  if (h < 3 || w < 3) return 0;                              /* lib.rs:411: 3x3 windows */
  /* gather every (own, neighbour) pair of an interior coloured centre with a coloured,
   * differently coloured 4-neighbour (lib.rs:414-434) */
  size_t n_raw = 0, raw_cap = 1024;
  uint64_t *raw = (uint64_t *)malloc(raw_cap * 2 * sizeof(uint64_t));
  if (!raw) return 0;
  for (size_t r = 1; r + 1 < h; ++r)
    for (size_t c = 1; c + 1 < w; ++c) {
      const uint64_t own = labels[r * w + c];
      if (own == WS_OR_UNCOLOURED) continue;
      for (int d = 0; d < 4; ++d) {
        const uint64_t q = labels[(r + N4_DR[d]) * w + (c + N4_DC[d])];
        if (q == WS_OR_UNCOLOURED || q == own) continue;
        if (n_raw == raw_cap) {
          raw_cap *= 2;
          uint64_t *t = (uint64_t *)realloc(raw, raw_cap * 2 * sizeof(uint64_t));
          if (!t) { free(raw); return 0; }
          raw = t;
        }
        raw[2 * n_raw] = own < q ? own : q;                  /* Merge is unordered: lib.rs:299-306 */
        raw[2 * n_raw + 1] = own < q ? q : own;
        ++n_raw;
      }
    }
  /* lib.rs:440-443: sort + dedup twice; only the resulting SET matters downstream */
  qsort(raw, n_raw, 2 * sizeof(uint64_t), cmp_pair);
  size_t n = 0;
  for (size_t i = 0; i < n_raw; ++i) {
    if (i && raw[2 * i] == raw[2 * i - 2] && raw[2 * i + 1] == raw[2 * i - 1]) continue;
    if (n < cap) { out_pairs[2 * n] = raw[2 * i]; out_pairs[2 * n + 1] = raw[2 * i + 1]; }
    ++n;
  }
  free(raw);
  return n;
}

/* ------------------------------------------------- make_colour_map (a7) -- */

typedef struct { uint64_t *v; size_t n, cap; } region_t;

static int region_has(const region_t *g, uint64_t x) {
  for (size_t i = 0; i < g->n; ++i) if (g->v[i] == x) return 1;
  return 0;
}
static int region_push(region_t *g, uint64_t x) {
  if (g->n == g->cap) {
    size_t nc = g->cap ? g->cap * 2 : 4;
    uint64_t *t = (uint64_t *)realloc(g->v, nc * sizeof(uint64_t));
    if (!t) return -1;
    g->v = t; g->cap = nc;
  }
  g->v[g->n++] = x;
  return 0;
}
static int cmp_u64(const void *a, const void *b) {
  const uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}

static int colour_map_faithful(uint64_t *base_map, size_t map_len, const uint64_t *pairs, size_t n_pairs) {
  region_t *regs = NULL;
  size_t n_regs = 0, regs_cap = 0;
  int rc = WS_OR_OK;
  for (size_t i = 0; i < n_pairs && rc == WS_OR_OK; ++i) {
    const uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
    long hit0 = -1, hit1 = -1;
    int duplicate = 0;
    for (size_t g = 0; g < n_regs; ++g) {                    /* lib.rs:488-503 */
      const int ha = region_has(&regs[g], a), hb = region_has(&regs[g], b);
      if (ha && hb) { duplicate = 1; break; }                /* lib.rs:489-492 */
      if (ha || hb) {
        if (hit0 < 0) hit0 = (long)g;
        else { hit1 = (long)g; break; }
      }
    }
    if (duplicate) continue;
    if (hit0 < 0) {                                          /* lib.rs:505-508: new region [a, b] */
      if (n_regs == regs_cap) {
        size_t nc = regs_cap ? regs_cap * 2 : 8;
        region_t *t = (region_t *)realloc(regs, nc * sizeof(region_t));
        if (!t) { rc = WS_OR_ERR_ALLOC; break; }
        regs = t; regs_cap = nc;
      }
      regs[n_regs].v = NULL; regs[n_regs].n = regs[n_regs].cap = 0;
      if (region_push(&regs[n_regs], a) || region_push(&regs[n_regs], b)) { rc = WS_OR_ERR_ALLOC; ++n_regs; break; }
      ++n_regs;
    } else if (hit1 < 0) {                                   /* lib.rs:509-514: extend, sort, dedup */
      region_t *g = &regs[hit0];
      if (region_push(g, a) || region_push(g, b)) { rc = WS_OR_ERR_ALLOC; break; }
      qsort(g->v, g->n, sizeof(uint64_t), cmp_u64);
      size_t m = 0;
      for (size_t k = 0; k < g->n; ++k) if (!k || g->v[k] != g->v[k - 1]) g->v[m++] = g->v[k];
      g->n = m;
    } else {                                                 /* lib.rs:515-532: lower-index region swallows the other */
      region_t *lo = &regs[hit0 < hit1 ? hit0 : hit1], *hi = &regs[hit0 < hit1 ? hit1 : hit0];
      for (size_t k = 0; k < hi->n; ++k) if (region_push(lo, hi->v[k])) { rc = WS_OR_ERR_ALLOC; break; }
      hi->n = 0;
    }
    size_t m = 0;                                            /* lib.rs:535: drop emptied regions, keep order */
    for (size_t g = 0; g < n_regs; ++g) {
      if (regs[g].n == 0) { free(regs[g].v); continue; }
      regs[m++] = regs[g];
    }
    n_regs = m;
  }
  if (rc == WS_OR_OK)
    for (size_t g = 0; g < n_regs; ++g) {                    /* lib.rs:538-541: value in region -> region[0] */
      const uint64_t rep = regs[g].v[0];
      for (size_t k = 0; k < map_len; ++k) if (region_has(&regs[g], base_map[k])) base_map[k] = rep;
    }
  for (size_t g = 0; g < n_regs; ++g) free(regs[g].v);
  free(regs);
  return rc;
}

static uint64_t uf_find(uint64_t *parent, uint64_t x) {
  uint64_t r = x;
  while (parent[r] != r) r = parent[r];
  while (parent[x] != r) { uint64_t nx = parent[x]; parent[x] = r; x = nx; }
  return r;
}

/* Same partition as the faithful closure; representative = smallest value of the class. */
static int colour_map_canonical(uint64_t *base_map, size_t map_len, const uint64_t *pairs, size_t n_pairs) {
  if (n_pairs == 0) return WS_OR_OK;
  /* values appearing in pairs are current map VALUES, i.e. < map_len for a sane map */
  uint64_t *parent = (uint64_t *)malloc(map_len * sizeof(uint64_t));
  if (!parent) return WS_OR_ERR_ALLOC;
  for (size_t k = 0; k < map_len; ++k) parent[k] = k;
  for (size_t i = 0; i < n_pairs; ++i) {
    uint64_t a = pairs[2 * i], b = pairs[2 * i + 1];
    if (a >= map_len || b >= map_len) continue;
    a = uf_find(parent, a); b = uf_find(parent, b);
    if (a == b) continue;
    if (a < b) parent[b] = a; else parent[a] = b;            /* min-root */
  }
  for (size_t k = 0; k < map_len; ++k)
    if (base_map[k] < map_len) base_map[k] = uf_find(parent, base_map[k]);
  free(parent);
  return WS_OR_OK;
}

int ws_or_make_colour_map(uint64_t *base_map, size_t map_len, const uint64_t *pairs, size_t n_pairs,
                          int map_mode) {
  return map_mode == WS_OR_MAP_FAITHFUL ? colour_map_faithful(base_map, map_len, pairs, n_pairs)
                                        : colour_map_canonical(base_map, map_len, pairs, n_pairs);
}

/* ------------------------------------------------ recolour / lake sizes -- */

void ws_or_recolour(uint64_t *labels, size_t n, const uint64_t *colour_map) {
  for (size_t i = 0; i < n; ++i) labels[i] = colour_map[labels[i]];   /* lib.rs:591 */
}

void ws_or_find_lake_sizes(const uint64_t *labels, size_t n, uint64_t *hist) {
  memset(hist, 0, (n + 1) * sizeof(uint64_t));                        /* lib.rs:630: len()+1 */
  for (size_t i = 0; i < n; ++i) hist[labels[i]]++;                   /* lib.rs:631-633 */
}

/* ------------------------------------------------------- merging (a9) -- */

int ws_or_merge(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc, size_t n_seeds,
                uint8_t max_water_level, int edge_correction, int tie_mode, uint64_t rng_seed,
                int map_mode, uint64_t *out_labels, ws_or_level_cb cb, void *user,
                ws_or_stats *stats) {
  plane_t pl;
  int rc = plane_setup(&pl, img, h, w, edge_correction);
  if (rc) return rc;
  const size_t n = pl.ph * pl.pw;
  if (stats) memset(stats, 0, sizeof *stats);
  memset(out_labels, 0, n * sizeof(uint64_t));               /* lib.rs:1337 */
  rc = paint_seeds(out_labels, pl.ph, pl.pw, seeds_rc, n_seeds);
  if (rc) { free(pl.pimg); return rc; }

  /* lib.rs:1360,1369: persistent colour map, identity, entry 0 = UNCOLOURED */
  const size_t map_len = n_seeds + 1;
  uint64_t *cmap = (uint64_t *)malloc(map_len * sizeof(uint64_t));
  uint64_t *buf_rc = (uint64_t *)malloc((n ? n : 1) * 2 * sizeof(uint64_t));
  uint64_t *buf_col = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint8_t *buf_cf = (uint8_t *)malloc(n ? n : 1);
  size_t pair_cap = 2 * n + 4;
  uint64_t *pairs = (uint64_t *)malloc(pair_cap * 2 * sizeof(uint64_t));
  if (!cmap || !buf_rc || !buf_col || !buf_cf || !pairs) {
    free(cmap); free(buf_rc); free(buf_col); free(buf_cf); free(pairs); free(pl.pimg);
    return WS_OR_ERR_ALLOC;
  }
  for (size_t k = 0; k < map_len; ++k) cmap[k] = k;
  uint64_t rng = rng_seed;

  for (unsigned lvl = 0; lvl <= max_water_level && rc == WS_OR_OK; ++lvl) {   /* lib.rs:1379 */
    colouring_loop(&pl, out_labels, (uint8_t)lvl, tie_mode, &rng, buf_rc, buf_col, buf_cf, NULL,
                   NULL, stats);
    const size_t np = ws_or_find_merge(out_labels, pl.ph, pl.pw, pairs, pair_cap);  /* lib.rs:1450 */
    if (stats) stats->merge_pairs += np;
    rc = ws_or_make_colour_map(cmap, map_len, pairs, np, map_mode);                /* lib.rs:1460 */
    if (np > 0) ws_or_recolour(out_labels, n, cmap);                               /* lib.rs:1464-1466 */
    if (cb) cb(user, (uint8_t)lvl, max_water_level, pl.img, out_labels, pl.ph, pl.pw);
  }
  free(cmap); free(buf_rc); free(buf_col); free(buf_cf); free(pairs); free(pl.pimg);
  return rc;
}

void ws_or_merge_transform_stub(size_t h, size_t w, uint64_t *out_labels) {
  memset(out_labels, 0, h * w * sizeof(uint64_t));           /* lib.rs:1529 */
  if (h < 2 || w < 2) return;                                /* (the reference's slice would panic) */
  for (size_t r = 1; r + 1 < h; ++r)
    for (size_t c = 1; c + 1 < w; ++c) out_labels[r * w + c] = 123;   /* lib.rs:1532 */
}

/* ------------------------------------------------ find_local_minima (a12) -- */

size_t ws_or_find_local_minima(const uint8_t *img, size_t h, size_t w, uint64_t *out_rc, size_t cap) {
  if (h < 3 || w < 3) return 0;
  size_t n = 0;
  for (size_t r = 1; r + 1 < h; ++r)
    for (size_t c = 1; c + 1 < w; ++c) {
      const uint8_t v = img[r * w + c];
      int ok = 1;
      for (int d = 0; d < 8 && ok; ++d)                      /* lib.rs:1190: all neighbours < centre */
        ok = img[(r + N8_DR[d]) * w + (c + N8_DC[d])] < v;
      if (!ok) continue;
      if (n < cap) { out_rc[2 * n] = r; out_rc[2 * n + 1] = c; }
      ++n;
    }
  return n;
}

/* ------------------------------------------------ reachable-sample check -- */

int ws_or_check_reachable(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds, uint8_t max_water_level, int edge_correction,
                          const uint64_t *cand, size_t *bad_index) {
  const size_t ph = h + (edge_correction ? 2 : 0), pw = w + (edge_correction ? 2 : 0);
  const size_t n = ph * pw;
  uint64_t *det = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  int32_t *al = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
  uint32_t *ar = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
  int verdict = 0;
  if (!det || !al || !ar) { free(det); free(al); free(ar); return WS_OR_ERR_ALLOC; }
  int rc = ws_or_segment(img, h, w, seeds_rc, n_seeds, max_water_level, edge_correction,
                         WS_OR_TIE_FIRST, 0, det, al, ar, NULL, NULL, NULL);
  if (rc) { free(det); free(al); free(ar); return rc; }
  /* arrival times and the coloured mask do not depend on the tie-break (lib.rs:224-231) */
  for (size_t p = 0; p < n && !verdict; ++p)
    if ((det[p] != 0) != (cand[p] != 0)) { verdict = 1; if (bad_index) *bad_index = p; }
  for (size_t p = 0; p < n && !verdict; ++p) {
    if (det[p] == 0) continue;
    if (al[p] == -1) {                                       /* seed pixel keeps its painted colour */
      if (cand[p] != det[p]) { verdict = 3; if (bad_index) *bad_index = p; }
      continue;
    }
    const size_t r = p / pw, c = p % pw;
    int ok = 0;
    for (int d = 0; d < 4 && !ok; ++d) {
      const size_t q = (r + N4_DR[d]) * pw + (c + N4_DC[d]);
      const int earlier = al[q] != -2 && (al[q] < al[p] || (al[q] == al[p] && ar[q] < ar[p]));
      ok = earlier && cand[q] == cand[p];
    }
    if (!ok) { verdict = 2; if (bad_index) *bad_index = p; }
  }
  free(det); free(al); free(ar);
  return verdict;
}

/* -------------------------------------------------------- canonicaliser -- */

size_t ws_or_canonicalise(uint64_t *labels, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds) {
  const size_t n = h * w;
  /* seed colour plane as painted by lib.rs:1365-1367 */
  uint64_t *seedcol = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
  /* label values are arbitrary (<= n_seeds for reference-shaped outputs); map value -> min seed colour */
  uint64_t maxlab = 0;
  for (size_t p = 0; p < n; ++p) if (labels[p] > maxlab) maxlab = labels[p];
  uint64_t *best = (uint64_t *)malloc((maxlab + 1) * sizeof(uint64_t));
  if (!seedcol || !best) { free(seedcol); free(best); return 0; }
  for (size_t i = 0; i < n_seeds; ++i) {
    const uint64_t r = seeds_rc[2 * i], c = seeds_rc[2 * i + 1];
    if (r < h && c < w) seedcol[r * w + c] = i + 1;
  }
  for (uint64_t v = 0; v <= maxlab; ++v) best[v] = UINT64_MAX;
  for (size_t p = 0; p < n; ++p)
    if (labels[p] && seedcol[p] && seedcol[p] < best[labels[p]]) best[labels[p]] = seedcol[p];
  size_t classes = 0;
  for (uint64_t v = 1; v <= maxlab; ++v) if (best[v] != UINT64_MAX) ++classes;
  for (size_t p = 0; p < n; ++p)
    if (labels[p] && best[labels[p]] != UINT64_MAX) labels[p] = best[labels[p]];
  free(seedcol); free(best);
  return classes;
}

/* ------------------------------------- arrival-time restatement (2nd form) -- */

typedef struct { uint64_t key; size_t p; } heap_item;
typedef struct { heap_item *a; size_t n, cap; } heap_t;

static int heap_push(heap_t *hp, uint64_t key, size_t p) {
  if (hp->n == hp->cap) {
    size_t nc = hp->cap ? hp->cap * 2 : 1024;
    heap_item *t = (heap_item *)realloc(hp->a, nc * sizeof(heap_item));
    if (!t) return -1;
    hp->a = t; hp->cap = nc;
  }
  size_t i = hp->n++;
  while (i) {
    size_t up = (i - 1) / 2;
    if (hp->a[up].key <= key) break;
    hp->a[i] = hp->a[up]; i = up;
  }
  hp->a[i].key = key; hp->a[i].p = p;
  return 0;
}
static heap_item heap_pop(heap_t *hp) {
  heap_item top = hp->a[0], last = hp->a[--hp->n];
  size_t i = 0;
  for (;;) {
    size_t l = 2 * i + 1, r = l + 1, m = i;
    uint64_t mk = last.key;
    if (l < hp->n && hp->a[l].key < mk) { m = l; mk = hp->a[l].key; }
    if (r < hp->n && hp->a[r].key < mk) { m = r; }
    if (m == i) break;
    hp->a[i] = hp->a[m]; i = m;
  }
  if (hp->n) hp->a[i] = last;
  return top;
}

#define KEY_INF UINT64_MAX
#define KEY(l, r) (((uint64_t)(l) << 32) | (uint64_t)(r))

/* arrival keys on the (padded) plane; key 0 = seed, KEY_INF = never coloured */
static int arrival_keys(const plane_t *pl, const uint64_t *seed_labels, uint8_t maxlvl, uint64_t *key) {
  const size_t ph = pl->ph, pw = pl->pw, n = ph * pw;
  heap_t hp = {NULL, 0, 0};
  for (size_t p = 0; p < n; ++p) {
    key[p] = seed_labels[p] ? 0 : KEY_INF;
    if (seed_labels[p] && heap_push(&hp, 0, p)) { free(hp.a); return WS_OR_ERR_ALLOC; }
  }
  while (hp.n) {
    heap_item it = heap_pop(&hp);
    if (it.key != key[it.p]) continue;
    const size_t r = it.p / pw, c = it.p % pw;
    for (int d = 0; d < 4; ++d) {
      const long rr = (long)r + N4_DR[d], cc = (long)c + N4_DC[d];
      /* only interior pixels can be flooded (lib.rs:220-222) */
      if (rr < 1 || cc < 1 || rr + 1 >= (long)ph || cc + 1 >= (long)pw) continue;
      const size_t q = (size_t)rr * pw + (size_t)cc;
      const unsigned iq = pl->img[q];
      if (iq > maxlvl) continue;                             /* never flooded (lib.rs:224) */
      /* coloured one ring after the neighbour, but not before its own level opens */
      uint64_t cand = it.key + 1;
      if (cand < KEY(iq, 1)) cand = KEY(iq, 1);
      if (cand < key[q]) {
        key[q] = cand;
        if (heap_push(&hp, cand, q)) { free(hp.a); return WS_OR_ERR_ALLOC; }
      }
    }
  }
  free(hp.a);
  return WS_OR_OK;
}

typedef struct { uint64_t key; size_t p; } order_item;
static int cmp_order(const void *a, const void *b) {
  const order_item *x = (const order_item *)a, *y = (const order_item *)b;
  if (x->key != y->key) return x->key < y->key ? -1 : 1;
  return x->p < y->p ? -1 : (x->p > y->p ? 1 : 0);
}

int ws_or_segment_arrival(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                          size_t n_seeds, uint8_t max_water_level, int edge_correction,
                          uint64_t *out_labels, uint64_t *arr_key) {
  plane_t pl;
  int rc = plane_setup(&pl, img, h, w, edge_correction);
  if (rc) return rc;
  const size_t n = pl.ph * pl.pw;
  memset(out_labels, 0, n * sizeof(uint64_t));
  rc = paint_seeds(out_labels, pl.ph, pl.pw, seeds_rc, n_seeds);
  uint64_t *key = arr_key ? arr_key : (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  order_item *ord = (order_item *)malloc((n ? n : 1) * sizeof(order_item));
  if (rc == WS_OR_OK && (!key || !ord)) rc = WS_OR_ERR_ALLOC;
  if (rc == WS_OR_OK) rc = arrival_keys(&pl, out_labels, max_water_level, key);
  if (rc == WS_OR_OK) {
    size_t m = 0;
    for (size_t p = 0; p < n; ++p)
      if (key[p] != 0 && key[p] != KEY_INF) { ord[m].key = key[p]; ord[m].p = p; ++m; }
    qsort(ord, m, sizeof(order_item), cmp_order);
    for (size_t i = 0; i < m; ++i) {                         /* parents arrive strictly earlier */
      const size_t p = ord[i].p, r = p / pl.pw, c = p % pl.pw;
      for (int d = 0; d < 4; ++d) {                          /* first in D,R,L,U coloured before p */
        const size_t q = (r + N4_DR[d]) * pl.pw + (c + N4_DC[d]);
        if (key[q] < key[p]) { out_labels[p] = out_labels[q]; break; }
      }
    }
  }
  if (!arr_key) free(key);
  free(ord); free(pl.pimg);
  return rc;
}

int ws_or_merge_arrival(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                        size_t n_seeds, uint8_t max_water_level, int edge_correction,
                        uint64_t *out_labels, ws_or_level_cb cb, void *user) {
  plane_t pl;
  int rc = plane_setup(&pl, img, h, w, edge_correction);
  if (rc) return rc;
  const size_t ph = pl.ph, pw = pl.pw, n = ph * pw;
  uint64_t *seedcol = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
  uint64_t *key = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint64_t *parent = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  uint64_t *minseed = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
  if (!seedcol || !key || !parent || !minseed) rc = WS_OR_ERR_ALLOC;
  if (rc == WS_OR_OK) rc = paint_seeds(seedcol, ph, pw, seeds_rc, n_seeds);
  if (rc == WS_OR_OK) rc = arrival_keys(&pl, seedcol, max_water_level, key);
  if (rc == WS_OR_OK) {
    for (size_t p = 0; p < n; ++p) { parent[p] = p; minseed[p] = seedcol[p] ? seedcol[p] : UINT64_MAX; }
    for (long lvl = -1; lvl <= (long)max_water_level; ++lvl) {
      /* activate the pixels of this level (seeds at -1) and join every adjacency the
       * reference can see: a 3x3 window centre is interior (lib.rs:411-434) */
      for (size_t p = 0; p < n; ++p) {
        const long pl_lvl = key[p] == 0 ? -1 : (key[p] == KEY_INF ? 1000 : (long)(key[p] >> 32));
        if (pl_lvl != lvl) continue;
        const size_t r = p / pw, c = p % pw;
        const int p_int = r >= 1 && c >= 1 && r + 1 < ph && c + 1 < pw;
        for (int d = 0; d < 4; ++d) {
          const long rr = (long)r + N4_DR[d], cc = (long)c + N4_DC[d];
          if (rr < 0 || cc < 0 || rr >= (long)ph || cc >= (long)pw) continue;
          const size_t q = (size_t)rr * pw + (size_t)cc;
          const long q_lvl = key[q] == 0 ? -1 : (key[q] == KEY_INF ? 1000 : (long)(key[q] >> 32));
          if (q_lvl > lvl) continue;
          const int q_int = rr >= 1 && cc >= 1 && rr + 1 < (long)ph && cc + 1 < (long)pw;
          if (!p_int && !q_int) continue;
          uint64_t a = uf_find(parent, p), b = uf_find(parent, q);
          if (a == b) continue;
          if (a < b) { parent[b] = a; if (minseed[b] < minseed[a]) minseed[a] = minseed[b]; }
          else { parent[a] = b; if (minseed[a] < minseed[b]) minseed[b] = minseed[a]; }
        }
      }
      if (lvl < 0) continue;
      for (size_t p = 0; p < n; ++p) {
        const int coloured = key[p] != KEY_INF && (key[p] == 0 || (long)(key[p] >> 32) <= lvl);
        out_labels[p] = coloured ? minseed[uf_find(parent, p)] : 0;
      }
      if (cb) cb(user, (uint8_t)lvl, max_water_level, pl.img, out_labels, ph, pw);
    }
  }
  free(seedcol); free(key); free(parent); free(minseed); free(pl.pimg);
  return rc;
}

/* ---------------------------------------------- pre_processor (lib.rs:1081-1173) -- */

#include <math.h>

static double pre_get(const void *data, int dtype, size_t i) {
  switch (dtype) {
    case 0: return (double)((const float *)data)[i];
    case 1: return ((const double *)data)[i];
    case 2: return (double)((const int32_t *)data)[i];
    case 3: return (double)((const uint16_t *)data)[i];
    case 4: return (double)((const int16_t *)data)[i];
    default: return (double)((const uint8_t *)data)[i];
  }
}

/* dtype: 0 f32, 1 f64, 2 i32, 3 u16, 4 i16, 5 u8.  Returns 0, or -1 when max_value is not in
 * 1..=254 (the reference asserts, lib.rs:1143-1144). */
int ws_or_pre_processor(const void *data, int dtype, size_t n, uint8_t max_value, uint8_t *out) {
  if (max_value >= WS_OR_NEVER_FILL || max_value <= WS_OR_ALWAYS_FILL) return -1;
  double mn = 0.0, mx = 0.0;                               /* lib.rs:1149, 1154: folds seeded with zero */
  for (size_t i = 0; i < n; ++i) {
    const double v = pre_get(data, dtype, i);
    if (v < mn && isfinite(v)) mn = v;                     /* lib.rs:1149 */
    if (v > mx && isfinite(v)) mx = v;                     /* lib.rs:1154 */
  }
  for (size_t i = 0; i < n; ++i) {
    const double v = pre_get(data, dtype, i);
    if (isnormal(v)) {                                     /* lib.rs:1161 */
      const double normal = (v - mn) / (mx - mn);          /* lib.rs:1163 */
      /* lib.rs:1164: `(normal * MAX).to_u8().unwrap()` truncates toward zero and PANICS outside 0..=255.  With the
       * zero-seeded folds mn <= 0 <= mx, and for finite v: mn <= v <= mx, so 0 <= normal <= 1 and the product lies
       * in [0, MAX] (MAX <= 254): the cast below is the reference's value wherever the reference returns one.  The only
       * inputs outside that are mx == mn (all values 0 / non-finite: no `normal` branch is taken) -- asserted, so that
       * an out-of-range product is an error here too instead of an undefined C conversion. */
      const double scaled = normal * (double)max_value;
      if (!(scaled >= 0.0 && scaled < 256.0)) return -2;    /* the reference would panic (unwrap on None) */
      out[i] = (uint8_t)scaled;
    } else if (isinf(v) && !signbit(v)) {
      out[i] = WS_OR_ALWAYS_FILL;                          /* lib.rs:1165-1167: +inf */
    } else {
      out[i] = WS_OR_NEVER_FILL;                           /* lib.rs:1168-1170: NaN, -inf, subnormal, zero */
    }
  }
  return 0;
}
