/*
 * ws_oracle_par.c -- the oracle's flood loop in the reference's rayon shape, for the
 * timed CPU baseline (bench.py "cpu_baseline", kind "port").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY -- see ws_oracle.h.
 *
 * Shape restated from the reference: a parallel scan of every 3x3 window that
 * collects the pixels to colour (lib.rs:220-256, rayon `into_par_iter().collect()`),
 * followed by a sequential scatter (lib.rs:1741-1743), repeated until the scan comes
 * back empty, for every water level (lib.rs:1689-1748).  Labels are 8-byte `usize`
 * like the reference's Array2<usize>.  Tie-break = col0 (lib.rs:245), so the result
 * is bit-identical to ws_or_segment(..., WS_OR_TIE_FIRST).
 *
 * OpenMP threads stand in for the rayon pool (tests/core_bench.rs:45-48 installs a
 * pool of N threads around the transform).  This is a port, not the reference: it
 * does not pay the reference's per-window Vec allocations (lib.rs:229, 237-242),
 * so it is, if anything, faster than the real crate on the same cores.
 */
#include "ws_oracle.h"

#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t *idx; uint64_t *col; size_t n, cap; } cand_buf;

static int cand_push(cand_buf *b, uint64_t idx, uint64_t col) {
  if (b->n == b->cap) {
    size_t nc = b->cap ? b->cap * 2 : 4096;
    uint64_t *ti = (uint64_t *)realloc(b->idx, nc * sizeof(uint64_t));
    if (!ti) return -1;
    b->idx = ti;
    uint64_t *tc = (uint64_t *)realloc(b->col, nc * sizeof(uint64_t));
    if (!tc) return -1;
    b->col = tc;
    b->cap = nc;
  }
  b->idx[b->n] = idx;
  b->col[b->n] = col;
  b->n++;
  return 0;
}

int ws_or_max_threads(void) { return omp_get_max_threads(); }

/* Segmenting transform, rayon-shaped.  threads <= 0 -> all OpenMP threads. */
int ws_or_segment_par(const uint8_t *img, size_t h, size_t w, const uint64_t *seeds_rc,
                      size_t n_seeds, uint8_t max_water_level, int threads, uint64_t *out_labels,
                      ws_or_stats *stats) {
  const size_t n = h * w;
  if (threads <= 0) threads = omp_get_max_threads();
  if (stats) memset(stats, 0, sizeof *stats);
  memset(out_labels, 0, n * sizeof(uint64_t));
  for (size_t i = 0; i < n_seeds; ++i) {
    const uint64_t r = seeds_rc[2 * i], c = seeds_rc[2 * i + 1];
    if (r >= h || c >= w) return WS_OR_ERR_SEED_OOB;
    out_labels[r * w + c] = (uint64_t)i + 1;
  }
  if (h < 3 || w < 3) return WS_OR_OK;
  cand_buf *bufs = (cand_buf *)calloc((size_t)threads, sizeof(cand_buf));
  if (!bufs) return WS_OR_ERR_ALLOC;
  int failed = 0;

  for (unsigned lvl = 0; lvl <= max_water_level && !failed; ++lvl) {
    uint64_t rings = 0;
    for (;;) {
      size_t total = 0;
#pragma omp parallel num_threads(threads) reduction(+ : total)
      {
        const int t = omp_get_thread_num(), nt = omp_get_num_threads();
        cand_buf *b = &bufs[t];
        b->n = 0;
        /* contiguous row blocks keep the collected order row-major */
        const size_t rows = h - 2;
        const size_t r0 = 1 + rows * (size_t)t / (size_t)nt, r1 = 1 + rows * (size_t)(t + 1) / (size_t)nt;
        for (size_t r = r0; r < r1; ++r) {
          const uint8_t *ir = img + r * w;
          const uint64_t *lr = out_labels + r * w;
          for (size_t c = 1; c + 1 < w; ++c) {
            if (ir[c] > lvl || lr[c] != 0) continue;
            const uint64_t d = lr[c + w], rt = lr[c + 1], lf = lr[c - 1], u = lr[c - w];
            const uint64_t pick = d ? d : (rt ? rt : (lf ? lf : u));   /* D,R,L,U */
            if (!pick) continue;
            if (cand_push(b, r * w + c, pick)) {
#pragma omp atomic write
              failed = 1;
            }
          }
        }
        total += b->n;
      }
      if (stats) stats->scans++;
      if (failed || total == 0) break;
      ++rings;
      for (int t = 0; t < threads; ++t)                       /* sequential scatter */
        for (size_t i = 0; i < bufs[t].n; ++i) out_labels[bufs[t].idx[i]] = bufs[t].col[i];
      if (stats) stats->flooded += total;
    }
    if (stats && rings > stats->max_rings) stats->max_rings = rings;
  }
  for (int t = 0; t < threads; ++t) { free(bufs[t].idx); free(bufs[t].col); }
  free(bufs);
  return failed ? WS_OR_ERR_ALLOC : WS_OR_OK;
}
