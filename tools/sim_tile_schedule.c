/* sim_tile_schedule.c -- CPU model of the relaxation's PASS SCHEDULE and ROUND RECIPE on long-range (smooth) maps: how many
 * passes, tile runs and rounds a tile geometry / a sequence of sweeps and scans needs, before any kernel is written for it.
 * Not product code, not the oracle: it solves the same fixpoint  key(p) = max(base(p), 1 + min4 key(q))  (ws_common.hpp)
 * by the engine's own scheme (ws_relax.hip) and counts.
 *
 *   - tiles of TW x TH pixels on one grid; a tile runs when a 4-neighbour changed a border pixel facing it in the pass
 *     before, or when it stopped at its round cap itself; every tile of a pass reads its halo as the pass found it;
 *   - a tile run = up to CAP rounds; a round = the operations of RECIPE in order, the last one checked (the run ends when
 *     it changes nothing):
 *       D R U L   patch sweeps (4 x 4 patches, Gauss-Seidel inside a patch in the sweep's direction, neighbour patches as
 *                 they were when the sweep started; rows of OTHER bands as they were at the last barrier)
 *       r l       exact row scans (right / left) over the whole tile width, from the tile's halo column
 *       d u       exact column scans (down / up) over the whole tile height, from the tile's halo row
 *       |         barrier: bands publish their boundary rows (the kernel has one before the checked sweep)
 *     the kernel's round is "DRrdUlu|L".
 *
 *   gcc -O2 -o tools/_build/sim_tile_schedule tools/sim_tile_schedule.c
 *   sim_tile_schedule image.u8 N seeds.u32 n_seeds TW TH CAP RECIPE [slots]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KEY_INF 0xFF000000u
#define PS 4

static uint32_t *key, *base, *snap;
static int N;
static long g_rounds, g_ops[128];
static long g_chg_hist[12], g_chg_runs, g_bbox_rows_hist[9], g_bbox_cols_hist[9];      /* SIM_CHANGED: changed pixels per tile run (log2 bins), height / width of their bounding box in eighths of the tile */
static int g_matters;
static int g_skip_l, g_skip_r;      /* columns at the left / right end of a tile whose changes do not count as a changed first / last row */
static uint32_t g_side_min[5]; /* smallest new key among the changed pixels of the top / bottom / left / right border, and of the tile */

static inline uint32_t med3(uint32_t lo, uint32_t x, uint32_t hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* tile-local copies: cur (live), beg (at sweep start), bar (at the last barrier); all (th + 2) x (tw + 2) with the halo ring */
static uint32_t *cur, *beg, *bar, *bs;
static int P; /* pitch */

static int relax_at(int x, int y, const uint32_t *up, const uint32_t *dn, const uint32_t *lf, const uint32_t *rt) {
  const int i = y * P + x;
  const uint32_t t = cur[i], b = bs[i];
  if (b >= t) return 0;
  uint32_t m = up[i - P];
  if (dn[i + P] < m) m = dn[i + P];
  if (lf[i - 1] < m) m = lf[i - 1];
  if (rt[i + 1] < m) m = rt[i + 1];
  const uint32_t n = med3(b, m + 1u, t);
  if (n == t) return 0;
  cur[i] = n;
  return 1;
}

/* one tile run; returns bit mask: 1 top row changed, 2 bottom, 4 left col, 8 right col, 16 anything, 32 stopped at cap */
static int run_tile(int x0, int y0, int tw, int th, int cap, const char *recipe) {
  P = tw + 2;
  const size_t cells = (size_t)(th + 2) * P;
  for (int y = -1; y <= th; ++y)
    for (int x = -1; x <= tw; ++x) {
      const int gy = y0 + y, gx = x0 + x, i = (y + 1) * P + x + 1;
      const int in = gy >= 0 && gy < N && gx >= 0 && gx < N;
      const int halo = y < 0 || y >= th || x < 0 || x >= tw;
      cur[i] = in ? (halo ? snap : key)[(size_t)gy * N + gx] : KEY_INF;
      bs[i] = in && !halo ? base[(size_t)gy * N + gx] : KEY_INF;
      if (bs[i] > cur[i]) bs[i] = cur[i];
    }
  memcpy(bar, cur, cells * 4);
  int res = 0, round = 0;
  const int nops = (int)strlen(recipe);
  for (;;) {
    int changed_last = 0;
    for (int k = 0; k < nops; ++k) {
      const char op = recipe[k];
      int changed = 0;
      if (op == '|') { memcpy(bar, cur, cells * 4); continue; }
      ++g_ops[(int)op];
      if (op == 'D' || op == 'R' || op == 'U' || op == 'L') {
        memcpy(beg, cur, cells * 4);
        for (int py = 0; py < th; py += PS)
          for (int px = 0; px < tw; px += PS)
            for (int a = 0; a < PS; ++a)
              for (int c = 0; c < PS; ++c) {
                /* D: rows top to bottom, pixels left to right; U: rows bottom to top; R: columns left to right, pixels top
                 * to bottom; L: columns right to left */
                const int yy = op == 'D' ? a : (op == 'U' ? PS - 1 - a : c);
                const int xx = op == 'R' ? a : (op == 'L' ? PS - 1 - a : c);
                const int x = px + xx + 1, y = py + yy + 1;
                changed |= relax_at(x, y, yy == 0 ? bar : cur, yy == PS - 1 ? bar : cur, xx == 0 ? beg : cur, xx == PS - 1 ? beg : cur);
              }
      } else if (op == 'r' || op == 'l') {
        for (int y = 1; y <= th; ++y)
          for (int k2 = 0; k2 < tw; ++k2) {
            const int x = op == 'r' ? 1 + k2 : tw - k2, i = y * P + x;
            const uint32_t t = cur[i], b = bs[i];
            if (b >= t) continue;
            const uint32_t n = med3(b, cur[op == 'r' ? i - 1 : i + 1] + 1u, t);
            if (n != t) { cur[i] = n; changed = 1; }
          }
      } else if (op == 'V' || op == 'A') {
        /* scanline pass: rows top to bottom (V) or bottom to top (A); a row first takes what the row before it offers,
         * then is scanned exactly in both directions: every path that never turns back vertically is settled */
        for (int k2 = 0; k2 < th; ++k2) {
          const int y = op == 'V' ? 1 + k2 : th - k2;
          for (int x = 1; x <= tw; ++x) {
            const int i = y * P + x;
            const uint32_t t = cur[i], b = bs[i];
            if (b >= t) continue;
            const uint32_t n = med3(b, cur[op == 'V' ? i - P : i + P] + 1u, t);
            if (n != t) { cur[i] = n; changed = 1; }
          }
          for (int dirx = 0; dirx < 2; ++dirx)
            for (int k3 = 0; k3 < tw; ++k3) {
              const int x = dirx == 0 ? 1 + k3 : tw - k3, i = y * P + x;
              const uint32_t t = cur[i], b = bs[i];
              if (b >= t) continue;
              const uint32_t n = med3(b, cur[dirx == 0 ? i - 1 : i + 1] + 1u, t);
              if (n != t) { cur[i] = n; changed = 1; }
            }
        }
      } else if (op == 'c') {
        /* both column scans in one phase: what enters a band from above / below comes through the OTHER bands as they were
         * when the phase started; inside the band first the rows downwards, then upwards */
        memcpy(beg, cur, cells * 4);
        for (int x = 1; x <= tw; ++x) {
          static uint32_t e_dn[4096], e_up[4096];
          uint32_t v = beg[0 * P + x];
          for (int y = 1; y <= th; ++y) { if ((y - 1) % PS == 0) e_dn[(y - 1) / PS] = v; const int i = y * P + x; v = med3(bs[i] > beg[i] ? beg[i] : bs[i], v + 1u, beg[i]); }
          v = beg[(th + 1) * P + x];
          for (int y = th; y >= 1; --y) { if (y % PS == 0) e_up[(y - 1) / PS] = v; const int i = y * P + x; v = med3(bs[i] > beg[i] ? beg[i] : bs[i], v + 1u, beg[i]); }
          for (int bnd = 0; bnd < th / PS; ++bnd) {
            uint32_t w = e_dn[bnd];
            for (int y = bnd * PS + 1; y <= bnd * PS + PS; ++y) { const int i = y * P + x; const uint32_t n = med3(bs[i], w + 1u, cur[i]); if (n != cur[i]) { cur[i] = n; changed = 1; } w = n; }
            w = e_up[bnd];
            for (int y = bnd * PS + PS; y >= bnd * PS + 1; --y) { const int i = y * P + x; const uint32_t n = med3(bs[i], w + 1u, cur[i]); if (n != cur[i]) { cur[i] = n; changed = 1; } w = n; }
          }
        }
      } else if (op == 'd' || op == 'u') {
        for (int x = 1; x <= tw; ++x)
          for (int k2 = 0; k2 < th; ++k2) {
            const int y = op == 'd' ? 1 + k2 : th - k2, i = y * P + x;
            const uint32_t t = cur[i], b = bs[i];
            if (b >= t) continue;
            const uint32_t n = med3(b, cur[op == 'd' ? i - P : i + P] + 1u, t);
            if (n != t) { cur[i] = n; changed = 1; }
          }
      }
      changed_last = changed;
    }
    ++round;
    ++g_rounds;
    if (!changed_last) break;
    if (cap && round >= cap) { res |= 32; break; }
  }
  for (int k = 0; k < 5; ++k) g_side_min[k] = 0xFFFFFFFFu;
  {
    long nchg = 0; int ymin = th, ymax = -1, xmin = tw, xmax = -1;
    for (int y = 0; y < th && y0 + y < N; ++y)
      for (int x = 0; x < tw && x0 + x < N; ++x)
        if (cur[(y + 1) * P + x + 1] != key[(size_t)(y0 + y) * N + x0 + x]) { ++nchg; if (y < ymin) ymin = y; if (y > ymax) ymax = y; if (x < xmin) xmin = x; if (x > xmax) xmax = x; }
    int b = 0; while ((1L << b) <= nchg && b < 11) ++b;
    ++g_chg_hist[b]; ++g_chg_runs;
    if (nchg) { ++g_bbox_rows_hist[((ymax - ymin + 1) * 8 + th - 1) / th]; ++g_bbox_cols_hist[((xmax - xmin + 1) * 8 + tw - 1) / tw]; }
  }
  for (int y = 0; y < th && y0 + y < N; ++y)
    for (int x = 0; x < tw && x0 + x < N; ++x) {
      const size_t p = (size_t)(y0 + y) * N + x0 + x;
      const uint32_t n = cur[(y + 1) * P + x + 1];
      if (n != key[p]) {
        key[p] = n;
        res |= 16;
        if (n < g_side_min[4]) g_side_min[4] = n;
        const int row_counts = x >= g_skip_l && x < tw - g_skip_r;
        /* SIM_MATTERS: a changed border pixel flags the tile across only if it can lower the pixel it touches there
         * (new + 1 < that pixel's stamp as this tile loaded it: the halo ring of cur[]); =2: and that pixel is not at its base */
        const int i2 = (y + 1) * P + x + 1;
        /* (the pixel across, in the plane: its base says whether it can fall at all) */
#define ACROSS_BASE(dx, dy) ((x0 + x + (dx) >= 0 && x0 + x + (dx) < N && y0 + y + (dy) >= 0 && y0 + y + (dy) < N) ? base[(size_t)(y0 + y + (dy)) * N + x0 + x + (dx)] : KEY_INF)
#define MATTERS2(off, dx, dy) (!g_matters || (n + 1u < cur[i2 + (off)] && (g_matters < 2 || (cur[i2 + (off)] > ACROSS_BASE(dx, dy) && n + 1u < cur[i2 + (off)]))))
#define MATTERS(off) MATTERS2(off, (off) == -1 ? -1 : ((off) == 1 ? 1 : 0), (off) == -P ? -1 : ((off) == P ? 1 : 0))
        if (y == 0 && row_counts && MATTERS(-P)) { res |= 1; if (n < g_side_min[0]) g_side_min[0] = n; }
        if ((y == th - 1 || y0 + y == N - 1) && row_counts && MATTERS(P)) { res |= 2; if (n < g_side_min[1]) g_side_min[1] = n; }
        if (x == 0 && MATTERS(-1)) { res |= 4; if (n < g_side_min[2]) g_side_min[2] = n; }
        if ((x == tw - 1 || x0 + x == N - 1) && MATTERS(1)) { res |= 8; if (n < g_side_min[3]) g_side_min[3] = n; }
      }
    }
  return res;
}

int main(int argc, char **argv) {
  if (argc < 9) { fprintf(stderr, "usage: image N seeds n_seeds TW TH CAP RECIPE [slots]\n"); return 2; }
  N = atoi(argv[2]);
  const size_t n = (size_t)N * N;
  uint8_t *img = malloc(n);
  FILE *f = fopen(argv[1], "rb");
  if (!f || fread(img, 1, n, f) != n) { fprintf(stderr, "image\n"); return 1; }
  fclose(f);
  const size_t ns = (size_t)atol(argv[4]);
  uint32_t *seeds = malloc(ns * 4 + 4);
  f = fopen(argv[3], "rb");
  if (!f || fread(seeds, 4, ns, f) != ns) { fprintf(stderr, "seeds\n"); return 1; }
  fclose(f);
  const int tw = atoi(argv[5]), th = atoi(argv[6]), cap = atoi(argv[7]);
  const char *recipe = argv[8];
  const int slots = argc > 9 ? atoi(argv[9]) : 512;
  key = malloc(n * 4);
  base = malloc(n * 4);
  snap = malloc(n * 4);
  const size_t cells = (size_t)(th + 2) * (tw + 2);
  cur = malloc(cells * 4); beg = malloc(cells * 4); bar = malloc(cells * 4); bs = malloc(cells * 4);
  for (size_t p = 0; p < n; ++p) {
    const int y = (int)(p / N), x = (int)(p % N);
    const int interior = y >= 1 && y < N - 1 && x >= 1 && x < N - 1;
    key[p] = KEY_INF;
    base[p] = interior && img[p] <= 254 ? ((uint32_t)img[p] << 24) | 1u : KEY_INF;
  }
  for (size_t i = 0; i < ns; ++i) { key[seeds[i]] = 0; base[seeds[i]] = 0; }
  const int tx = (N + tw - 1) / tw, ty = (N + th - 1) / th;
  /* pending[t]: smallest key that is waiting to enter tile t (0xFFFFFFFF: nothing pending).  SIM_DELTA (in ring units; L<n>
   * = n levels): a pass only runs the tiles whose pending key is within delta of the smallest pending key -- tiles further
   * up the flood order wait until what reaches them is (more nearly) final. */
  uint32_t *pending = malloc((size_t)tx * ty * 4), *pnext = malloc((size_t)tx * ty * 4);
  for (size_t t = 0; t < (size_t)tx * ty; ++t) pending[t] = 0;
  uint64_t delta = 0;
  if (getenv("SIM_DELTA")) { const char *e = getenv("SIM_DELTA"); delta = e[0] == 'L' ? (uint64_t)atol(e + 1) << 24 : (uint64_t)atol(e); }
  /* SIM_ADAPT=lo,hi: delta doubles after a pass that ran fewer than lo tiles while others waited, halves after one that ran more than hi */
  long adapt_lo = 0, adapt_hi = 0;
  if (getenv("SIM_ADAPT")) sscanf(getenv("SIM_ADAPT"), "%ld,%ld", &adapt_lo, &adapt_hi);
  long passes = 0, runs = 0, waves = 0;
  const int verbose = getenv("SIM_VERBOSE") != NULL;
  g_matters = getenv("SIM_MATTERS") ? atoi(getenv("SIM_MATTERS")) : 0;
  if (getenv("SIM_QUEUE")) {
    /* SIM_QUEUE=fifo|prio: no passes at all -- SIM_WORKERS (512) workers take tiles off a queue; a finished run queues the
     * neighbours whose border it changed (a neighbour that is running is marked and queues itself when it ends).  prio: the
     * worker takes the queued tile with the smallest pending stamp >> SIM_PRIO_SHIFT (24: its level), oldest first among
     * equals.  A run costs SIM_COST="fixed,per_round" microseconds (8,3); its result is applied when it starts (a little
     * optimistic: the kernel publishes at the end), its flags are delivered when it ends. */
    const int prio = strcmp(getenv("SIM_QUEUE"), "prio") == 0;
    const int W = getenv("SIM_WORKERS") ? atoi(getenv("SIM_WORKERS")) : 512;
    const int shift = getenv("SIM_PRIO_SHIFT") ? atoi(getenv("SIM_PRIO_SHIFT")) : 24;
    const int tie_rand = getenv("SIM_TIE") && strcmp(getenv("SIM_TIE"), "rand") == 0;
    double c_fixed = 8.0, c_round = 3.0;
    if (getenv("SIM_COST")) sscanf(getenv("SIM_COST"), "%lf,%lf", &c_fixed, &c_round);
    const size_t T = (size_t)tx * ty;
    uint8_t *state = calloc(T, 1);          /* 0 idle, 1 queued, 2 running, 3 running and flagged again */
    uint64_t *seq = calloc(T, 8);
    uint64_t next_seq = 1;
    struct Run { double end; size_t t; int r; uint32_t side[5]; } *run = calloc((size_t)W, sizeof *run);
    int busy = 0;
    for (size_t t = 0; t < T; ++t) { state[t] = 1; seq[t] = next_seq++; pending[t] = 0; }
    free(snap); snap = key;      /* a run reads the plane as it is when it starts */
    double now = 0.0, work = 0.0;
    long queued = (long)T;
    for (;;) {
      while (busy < W && queued > 0) {
        size_t best = T;
        for (size_t t = 0; t < T; ++t) {
          if (state[t] != 1) continue;
          if (best == T) { best = t; continue; }
          const uint32_t a = prio ? pending[t] >> shift : 0, b = prio ? pending[best] >> shift : 0;
          /* SIM_TIE=rand: among equals any tile (a bitmap per bucket has no order), else the one queued first */
          const uint64_t sa = tie_rand ? (seq[t] * 0x9E3779B97F4A7C15ull) >> 20 : seq[t], sb = tie_rand ? (seq[best] * 0x9E3779B97F4A7C15ull) >> 20 : seq[best];
          if (a < b || (a == b && sa < sb)) best = t;
        }
        const long r0 = g_rounds;
        const int i = (int)(best % tx), j = (int)(best / tx);
        state[best] = 2; pending[best] = 0xFFFFFFFFu; --queued;
        const int r = run_tile(i * tw, j * th, tw, th, cap, recipe);
        const double c = c_fixed + c_round * (double)(g_rounds - r0);
        work += c; ++runs;
        run[busy].end = now + c; run[busy].t = best; run[busy].r = r;
        memcpy(run[busy].side, g_side_min, sizeof g_side_min);
        ++busy;
      }
      if (!busy) break;
      int k = 0;
      for (int q = 1; q < busy; ++q) if (run[q].end < run[k].end) k = q;
      const struct Run e = run[k];
      run[k] = run[--busy];
      now = e.end;
      const size_t t = e.t;
      const int i = (int)(t % tx), j = (int)(t / tx);
#define NOTE(tt, v) do { if ((v) < pending[tt]) pending[tt] = (v); if (state[tt] == 0) { state[tt] = 1; seq[tt] = next_seq++; ++queued; } else if (state[tt] == 2) state[tt] = 3; } while (0)
      if ((e.r & 1) && j > 0) NOTE(t - tx, e.side[0]);
      if ((e.r & 2) && j + 1 < ty) NOTE(t + tx, e.side[1]);
      if ((e.r & 4) && i > 0) NOTE(t - 1, e.side[2]);
      if ((e.r & 8) && i + 1 < tx) NOTE(t + 1, e.side[3]);
      if (state[t] == 3 || (e.r & 32)) { if ((e.r & 32) && e.side[4] < pending[t]) pending[t] = e.side[4]; state[t] = 1; seq[t] = next_seq++; ++queued; }
      else state[t] = 0;
    }
    uint64_t sum = 0;
    for (size_t p = 0; p < n; ++p) sum += key[p] * (uint64_t)(p % 1000003 + 1);
    printf("queue %-4s workers %d tile %d x %d cap %d %s: tile runs %ld (%.1f per tile) rounds %ld  makespan %.0f us  work %.0f us (%.0f %% of the workers' time)  checksum %llx\n",
           prio ? "prio" : "fifo", W, tw, th, cap, recipe, runs, (double)runs / (double)T, g_rounds, now, work, 100.0 * work / (now * W), (unsigned long long)sum);
    return 0;
  }
  for (;;) {
    long ran = 0, waiting = 0;
    const long r0 = g_rounds;
    uint32_t lowest = 0xFFFFFFFFu;
    for (size_t t = 0; t < (size_t)tx * ty; ++t) if (pending[t] < lowest) lowest = pending[t];
    if (lowest == 0xFFFFFFFFu) break;
    const uint64_t limit = delta && passes > 0 ? (uint64_t)lowest + delta : 0xFFFFFFFEull;
    for (size_t t = 0; t < (size_t)tx * ty; ++t) pnext[t] = pending[t] != 0xFFFFFFFFu && pending[t] <= limit ? 0xFFFFFFFFu : pending[t];
    memcpy(snap, key, n * 4);
    for (int j = 0; j < ty; ++j)
      for (int i = 0; i < tx; ++i) {
        const size_t t = (size_t)j * tx + i;
        if (pending[t] != 0xFFFFFFFFu && pending[t] > limit) ++waiting;
        if (pending[t] == 0xFFFFFFFFu || pending[t] > limit) continue;
        ++ran;
        const int r = run_tile(i * tw, j * th, tw, th, cap, recipe);
#define RAISE(tt, v) do { if ((v) < pnext[tt]) pnext[tt] = (v); } while (0)
        if (r & 32) RAISE(t, g_side_min[4]);
        if ((r & 1) && j > 0) RAISE(t - tx, g_side_min[0]);
        if ((r & 2) && j + 1 < ty) RAISE(t + tx, g_side_min[1]);
        if ((r & 4) && i > 0) RAISE(t - 1, g_side_min[2]);
        if ((r & 8) && i + 1 < tx) RAISE(t + 1, g_side_min[3]);
      }
    if (verbose) printf("  pass %ld: tiles %ld rounds %ld lowest %08x\n", passes, ran, g_rounds - r0, lowest);
    if (getenv("SIM_CHANGED") && passes == atol(getenv("SIM_CHANGED"))) { memset(g_chg_hist, 0, sizeof g_chg_hist); memset(g_bbox_rows_hist, 0, sizeof g_bbox_rows_hist); memset(g_bbox_cols_hist, 0, sizeof g_bbox_cols_hist); g_chg_runs = 0; }
    if (adapt_hi && passes > 0) {
      if (ran < adapt_lo && waiting && delta < (1ull << 31)) delta *= 2;
      else if (ran > adapt_hi && delta > 16) delta /= 2;
    }
    ++passes;
    runs += ran;
    waves += (ran + slots - 1) / slots;
    uint32_t *tt = pending; pending = pnext; pnext = tt;
  }
  if (getenv("SIM_REPAIR")) {
    /* The seam repair of ws_relax.hip, replayed: pass 0 (every tile to its own fixpoint), then bands of +-kb rows along the
     * horizontal seams (tiles tw wide), then strips of +-ks columns along the vertical seams (slices th tall), every launch
     * with the halo as the launch found it; which regular tiles get flagged, and why.  SIM_REPAIR="kb,ks[,order]" with order
     * 0 = bands then strips (the engine), 1 = strips then bands, 2 = bands, strips, then a second strip launch half a
     * slice higher, 3 = as 2 plus a second band launch half a tile to the right. */
    int kb = 4, ks = 4, order = 0;
    sscanf(getenv("SIM_REPAIR"), "%d,%d,%d", &kb, &ks, &order);
    uint32_t *truth = malloc(n * 4);
    memcpy(truth, key, n * 4);
    for (size_t p = 0; p < n; ++p) key[p] = KEY_INF;
    for (size_t i = 0; i < ns; ++i) key[seeds[i]] = 0;
    memcpy(snap, key, n * 4);
    for (int j = 0; j < ty; ++j)
      for (int i = 0; i < tx; ++i) run_tile(i * tw, j * th, tw, th, 0, recipe);
    uint8_t *flag = calloc((size_t)tx * ty, 1);
    long why[4] = {0, 0, 0, 0};      /* band rows, strip columns, strip slice rows, band columns (only when no strips follow) */
    /* scratch tiles for other shapes */
    free(cur); free(beg); free(bar); free(bs);
    const size_t big = (size_t)(th + 2 * kb + 2 + 64) * (tw + 2 * ks + 2 + 64);
    cur = malloc(big * 4); beg = malloc(big * 4); bar = malloc(big * 4); bs = malloc(big * 4);
#define FLAG(i_, j_, w_) do { if ((i_) >= 0 && (i_) < tx && (j_) >= 0 && (j_) < ty) { if (!flag[(size_t)(j_) * tx + (i_)]) ++why[w_]; flag[(size_t)(j_) * tx + (i_)] = 1; } } while (0)
    for (int phase = 0; phase < 4; ++phase) {
      int what;      /* 0 bands, 1 strips, 2 strips shifted, 3 bands shifted, -1 nothing */
      if (order == 0) what = phase == 0 ? 0 : (phase == 1 ? 1 : -1);
      else if (order == 1) what = phase == 0 ? 1 : (phase == 1 ? 0 : -1);
      else if (order == 2) what = phase == 0 ? 0 : (phase == 1 ? 1 : (phase == 2 ? 2 : -1));
      else what = phase == 0 ? 0 : (phase == 1 ? 1 : (phase == 2 ? 2 : 3));
      if (what < 0) continue;
      const int last_bands = (order == 1);      /* no strips after the bands: their columns flag too */
      memcpy(snap, key, n * 4);
      if (what == 0 || what == 3) {
        const int xoff = what == 3 ? tw / 2 : 0;
        for (int j = 1; j < ty; ++j)
          for (int i = 0; i * tw - xoff < N; ++i) {
            const int x0 = i * tw - xoff;
            /* what a band changes in the columns a strip will look at again is the strip's business */
            if (getenv("SIM_REPAIR_SKIP") && !last_bands && what == 0) { g_skip_l = x0 > 0 ? ks : 0; g_skip_r = x0 + tw < N ? ks : 0; }
            const int r = run_tile(x0 < 0 ? 0 : x0, j * th - kb, x0 < 0 ? tw + x0 : tw, 2 * kb, 0, recipe);
            g_skip_l = g_skip_r = 0;
            const int ti = (x0 < 0 ? 0 : x0) / tw;
            if (r & 1) { FLAG(ti, j - 1, 0); if (xoff) FLAG(ti + 1, j - 1, 0); }
            if (r & 2) { FLAG(ti, j, 0); if (xoff) FLAG(ti + 1, j, 0); }
            if ((last_bands || what == 3) && (r & 4)) { FLAG((x0 - 1) / tw, j - 1, 3); FLAG((x0 - 1) / tw, j, 3); }
            if ((last_bands || what == 3) && (r & 8)) { FLAG((x0 + tw) / tw, j - 1, 3); FLAG((x0 + tw) / tw, j, 3); }
          }
      } else {
        const int yoff = what == 2 ? th / 2 : 0;
        for (int i = 1; i < tx; ++i)
          for (int j = 0; j * th - yoff < N; ++j) {
            const int y0 = j * th - yoff;
            const int r = run_tile(i * tw - ks, y0 < 0 ? 0 : y0, 2 * ks, y0 < 0 ? th + y0 : th, 0, recipe);
            const int tj = (y0 < 0 ? 0 : y0) / th;
            if (r & 4) { FLAG(i - 1, tj, 1); if (yoff) FLAG(i - 1, tj + 1, 1); }
            if (r & 8) { FLAG(i, tj, 1); if (yoff) FLAG(i, tj + 1, 1); }
            /* a slice's first / last row: covered by what follows? bands after strips cover the rows at the seams */
            const int rows_covered = (order == 1 && what == 1) || (order >= 2 && what == 1);
            if (!rows_covered && (r & 1)) { FLAG(i - 1, (y0 - 1) / th, 2); FLAG(i, (y0 - 1) / th, 2); }
            if (!rows_covered && (r & 2)) { FLAG(i - 1, (y0 + th) / th, 2); FLAG(i, (y0 + th) / th, 2); }
          }
      }
    }
    long flagged = 0, wrong = 0, wrong_unflagged = 0;
    for (size_t t = 0; t < (size_t)tx * ty; ++t) flagged += flag[t];
    for (int y = 0; y < N; ++y)
      for (int x = 0; x < N; ++x)
        if (key[(size_t)y * N + x] != truth[(size_t)y * N + x]) { ++wrong; if (!flag[(size_t)(y / th) * tx + x / tw]) ++wrong_unflagged; }
    printf("repair kb %d ks %d order %d: %ld of %d tiles flagged (%.1f %%): first flagged by band rows %ld, strip columns %ld, strip slice rows %ld, band columns %ld; "
           "%ld stamps still wrong, %ld of them in tiles nobody flagged (those need a neighbour's flag: an equation next to them is violated)\n",
           kb, ks, order, flagged, tx * ty, 100.0 * flagged / (tx * ty), why[0], why[1], why[2], why[3], wrong, wrong_unflagged);
    memcpy(key, truth, n * 4);
  }
  if (getenv("SIM_SEAM_REPORT")) {
    /* key[] is the fixpoint now.  Replay pass 0 alone and report how far from the nearest tile seam the pixels are that it
     * leaves wrong: what a repair pass confined to bands along the seams would have to reach. */
    uint32_t *truth = malloc(n * 4);
    memcpy(truth, key, n * 4);
    for (size_t p = 0; p < n; ++p) key[p] = KEY_INF;
    for (size_t i = 0; i < ns; ++i) key[seeds[i]] = 0;
    memcpy(snap, key, n * 4);
    for (int j = 0; j < ty; ++j)
      for (int i = 0; i < tx; ++i) run_tile(i * tw, j * th, tw, th, cap, recipe);
    long hist[64] = {0}, wrong = 0;
    for (int y = 0; y < N; ++y)
      for (int x = 0; x < N; ++x) {
        const size_t p = (size_t)y * N + x;
        if (key[p] == truth[p]) continue;
        ++wrong;
        int dx = x % tw, dy = y % th;
        dx = dx < tw - 1 - dx ? dx : tw - 1 - dx;
        dy = dy < th - 1 - dy ? dy : th - 1 - dy;
        const int d = dx < dy ? dx : dy;
        ++hist[d < 63 ? d : 63];
      }
    printf("after pass 0 alone (cap %d, %s): %ld of %zu pixels not final (%.3f %%); by distance to the nearest seam:", cap, recipe, wrong, n, 100.0 * wrong / n);
    long cum = 0;
    for (int d = 0; d < 64; ++d) { cum += hist[d]; if (hist[d]) printf(" %d:%ld", d, hist[d]); }
    printf("\n");
    memcpy(key, truth, n * 4);
  }
  if (getenv("SIM_CHANGED")) {
    printf("tile runs after pass %s: %ld; changed pixels per run, bins 0, 1, 2-3, 4-7, ... >=1024:", getenv("SIM_CHANGED"), g_chg_runs);
    for (int b = 0; b < 12; ++b) printf(" %ld", g_chg_hist[b]);
    printf("\n  bounding box of the changes, rows in eighths of the tile height (1..8):");
    for (int b = 1; b <= 8; ++b) printf(" %ld", g_bbox_rows_hist[b]);
    printf("\n  columns in eighths of the tile width:");
    for (int b = 1; b <= 8; ++b) printf(" %ld", g_bbox_cols_hist[b]);
    printf("\n");
  }
  uint64_t sum = 0;
  for (size_t p = 0; p < n; ++p) sum += key[p] * (uint64_t)(p % 1000003 + 1);
  /* vector instructions per wave (ws_relax.hip, ISA counts): a patch sweep ~90, the row scans of a wave's 4 rows ~220, a
   * column scan ~100; per tile run the load / write-back / flag epilogue ~300 */
  const double cost = 90.0 * (g_ops['D'] + g_ops['R'] + g_ops['U'] + g_ops['L']) + 220.0 * (g_ops['r'] + g_ops['l']) + 100.0 * (g_ops['d'] + g_ops['u']) + 120.0 * g_ops['c'] + 480.0 * (g_ops['V'] + g_ops['A']) + 300.0 * runs;
  printf("cost %7.1f M  ", cost / 1e6);
  printf("tile %4d x %4d cap %d %-12s: passes %5ld  tile runs %8ld (%.1f per tile)  rounds %8ld (%.1f per tile)  slot waves %ld  checksum %llx\n", tw, th,
         cap, recipe, passes, runs, (double)runs / ((double)tx * ty), g_rounds, (double)g_rounds / ((double)tx * ty), waves, (unsigned long long)sum);
  return 0;
}
