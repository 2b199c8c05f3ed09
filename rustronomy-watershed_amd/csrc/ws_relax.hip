// ws_relax.hip -- relaxation kernel of the fused engine (gfx950, wave64).
//
// Fixpoint: key(p) = max(base(p), 1 + min over the 4 neighbours of key(q))   (ws_common.hpp).
//
// Layout
//   * a lane owns a 4 x 4 patch of pixels IN REGISTERS (16 stamps + 16 bases);
//   * a wave is a 256-pixel-wide band (64 lanes x 4 columns); the left/right neighbour columns come
//     from the adjacent LANES with one DPP wave shift each (v_mov_b32_dpp wave_shr:1 / wave_shl:1;
//     the `old` operand supplies the tile's halo column for lane 0 / lane 63): no LDS, no conflicts;
//   * the NW waves of a workgroup are NW bands stacked vertically: tile = 256 x 4*NW pixels; only
//     band boundary rows go through LDS, one ds_write_b128 / ds_read_b128 per lane and row;
//   * global loads/stores are 16 B per lane, 1 KiB per wave instruction, row contiguous, and every
//     load is unconditional on a clamped address (a load under a data-dependent branch is waited
//     for before the next is issued -- that serialised ~50 round trips per thread in the first
//     version of this engine);
//   * one iteration = four sweeps (down, right, up, left); a sweep updates a whole patch row (or
//     column) at a time, i.e. 4 independent pixel updates, so dependency chains are 4 long;
//   * TWO tilings alternate: even passes use the grid anchored at (0, 0), odd passes the grid shifted
//     by half a tile (128, 2*NW), so the borders of one pass lie in the middle of the next pass's
//     tiles.  On the bench field 99.3 % of the stamps are final after pass 0 and every wrong one lies
//     within 8 px of a tile border (tools/exp_apron.py): the shifted pass repairs them with correct
//     surroundings, instead of the 3-4 passes it takes to walk a correction back and forth over the
//     same border.  A tile runs in pass k iff a tile of pass k-1 changed one of ITS border pixels
//     inside the quadrant the two tiles share: a pixel equation can only be left violated next to a
//     border pixel that a neighbour changed after it was read, and that border pixel sits in the
//     interior (or halo) of exactly the tile of the other grid that owns the pixel.
//
// What the measurements said (tools/diag_relax.hip, per-workgroup s_memrealtime stamps, MI355X):
// with ~5 global atomics per tile on shared words (statistics + convergence counters) a full pass
// took 440 us however the sweeps were organised: same-address atomics retire at ~12 ns each, and
// 8192 tiles x 4-5 of them IS 440 us.  Hence: no same-address atomic on the tile path.  Convergence
// words are plain stores of 1 into striped slots (idempotent), statistics are striped counters
// that exist only when profiling is on.
#include "ws_common.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>

namespace wsk {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// Diagnostic build only (tools/diag_relax.hip defines WS_DIAG_STAMPS): per-workgroup phase stamps.
#ifdef WS_DIAG_STAMPS
__device__ unsigned long long *g_diag = nullptr;
#define WS_STAMP(slot)                                                                          \
  do {                                                                                          \
    if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#define WS_STAMP_VALUE(slot, v)                                                                 \
  do {                                                                                          \
    if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + (slot)] = (v);              \
  } while (0)
// time spent between two points, summed over a tile run (thread 0's view): WS_ACC_T0 / WS_ACC(slot) pairs
#define WS_ACC_DECL unsigned long long ws_acc_t = 0, ws_acc[3] = {0, 0, 0}
#define WS_ACC_T0 (ws_acc_t = __builtin_amdgcn_s_memrealtime())
#define WS_ACC(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); ws_acc[k] += n_ - ws_acc_t; ws_acc_t = n_; } while (0)
#define WS_ACC_STORE do { if (threadIdx.x == 0 && g_diag) { g_diag[(size_t)blockIdx.x * 8 + 5] = ws_acc[0]; g_diag[(size_t)blockIdx.x * 8 + 6] = ws_acc[1]; g_diag[(size_t)blockIdx.x * 8 + 7] = ws_acc[2]; } } while (0)
#else
#define WS_ACC_DECL do {} while (0)
#define WS_ACC_T0 do {} while (0)
#define WS_ACC(k) do {} while (0)
#define WS_ACC_STORE do {} while (0)
#define WS_STAMP(slot) do {} while (0)
#define WS_STAMP_VALUE(slot, v) do {} while (0)
#endif

constexpr int RX_TW = 256;   // tile width: 64 lanes x 4 columns
constexpr int RX_P = 4;      // patch side

__device__ __forceinline__ uint32_t lane_left(uint32_t old, uint32_t v) {     // lane i <- lane i-1, lane 0 keeps old
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t lane_right(uint32_t old, uint32_t v) {    // lane i <- lane i+1, lane 63 keeps old
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x130, 0xF, 0xF, false);
}

// one pixel: key <- min(key, max(base, 1 + min4)).  The kernel keeps b <= t for every pixel (pixels
// that can never change -- seeds, the image border, halo copies -- carry b = t), and under b <= t
// min(t, max(b, x)) is the median of (b, x, t): v_min_u32, v_min3_u32, v_add_u32, v_med3_u32.
__device__ __forceinline__ uint32_t med3u(uint32_t a, uint32_t b, uint32_t c) {
  return max(min(a, b), min(max(a, b), c));
}
template <bool TRACK>
__device__ __forceinline__ void relax_px(uint32_t &t, uint32_t b, uint32_t u, uint32_t d, uint32_t l, uint32_t r, bool &changed) {
  const uint32_t n = med3u(b, min(min(u, d), min(l, r)) + 1u, t);
  if (TRACK) changed |= n != t;      // v_cmp + a scalar OR: the flag lives in an SGPR pair
  t = n;
}

// element p of a u32 plane, or bit p of a bit plane
__device__ __forceinline__ uint32_t plane_or_bit(const uint32_t *src, size_t p, int bits) {
  return bits ? (src[p >> 5] >> (p & 31u)) & 1u : src[p];
}

typedef uint32_t patch_t[RX_P][RX_P];

// The image bytes of a patch (one dword per row) -> the pixels' bases: (level << 24) | 1, or KEY_INF for a level that never
// opens.  With the default maximum level, 254 (lib.rs:942), only byte 255 never opens, and (255 << 24) | 1 lies ABOVE every
// stamp: the `b = min(b, t)` that follows every load pins such a pixel at its stamp by itself -- no compare, no select, and
// the byte comes into place with one shift and one and-or (7 cycles per pixel instead of 16.5: these kernels are bound by
// vector issue, profiles/r3_v0_issue_counters.json).  Called AFTER the loop that loads the rows, with its one (kernel
// uniform) branch outside the row loop: a branch between two rows' loads makes every row a memory round trip of its own
// (pass 0: 168 -> 189 us, measured).
__device__ __forceinline__ void patch_bases(const uint32_t (&iv)[RX_P], patch_t &B, uint32_t max_level) {
  if (max_level == 254u) {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      B[r][0] = (iv[r] << 24) | 1u;
      B[r][1] = ((iv[r] << 16) & 0xFF000000u) | 1u;
      B[r][2] = ((iv[r] << 8) & 0xFF000000u) | 1u;
      B[r][3] = (iv[r] & 0xFF000000u) | 1u;
    }
  } else {
#pragma unroll
    for (int r = 0; r < RX_P; ++r)
#pragma unroll
      for (int c = 0; c < RX_P; ++c) {
        const uint32_t v = (iv[r] >> (8 * c)) & 0xFFu;
        B[r][c] = v <= max_level ? ((v << 24) | 1u) : KEY_INF;
      }
  }
}

// A sweep walks the patch rows (or columns) in its direction and, inside a row, the pixels left to right
// (top to bottom), every pixel seeing its neighbours as they are NOW -- Gauss-Seidel all the way.  (Any
// order is a valid relaxation; taking a row's "old" left/right values instead cost 40 register copies
// per round.)
template <bool TRACK, bool DOWN>
__device__ __forceinline__ void sweep_rows(patch_t &T, const patch_t &B, const uint32_t (&up)[RX_P], const uint32_t (&dn)[RX_P],
                                           const uint32_t (&L)[RX_P], const uint32_t (&R)[RX_P], bool &changed) {
#pragma unroll
  for (int k = 0; k < RX_P; ++k) {
    const int r = DOWN ? k : RX_P - 1 - k;
#pragma unroll
    for (int c = 0; c < RX_P; ++c)
      relax_px<TRACK>(T[r][c], B[r][c], r == 0 ? up[c] : T[r - 1][c], r == RX_P - 1 ? dn[c] : T[r + 1][c],
                      c == 0 ? L[r] : T[r][c - 1], c == RX_P - 1 ? R[r] : T[r][c + 1], changed);
  }
}
template <bool TRACK, bool RIGHT>
__device__ __forceinline__ void sweep_cols(patch_t &T, const patch_t &B, const uint32_t (&up)[RX_P], const uint32_t (&dn)[RX_P],
                                           const uint32_t (&L)[RX_P], const uint32_t (&R)[RX_P], bool &changed) {
#pragma unroll
  for (int k = 0; k < RX_P; ++k) {
    const int c = RIGHT ? k : RX_P - 1 - k;
#pragma unroll
    for (int r = 0; r < RX_P; ++r)
      relax_px<TRACK>(T[r][c], B[r][c], r == 0 ? up[c] : T[r - 1][c], r == RX_P - 1 ? dn[c] : T[r + 1][c],
                      c == 0 ? L[r] : T[r][c - 1], c == RX_P - 1 ? R[r] : T[r][c + 1], changed);
  }
}

// Which tiles of the chunk starting at `first` have to run in this pass?  Lane k answers for tile
// first + k; the ballot is the to-do list.  tilesX x tilesY is this pass's grid, otherX x otherY the
// grid of the previous pass.
// Stamp words of the SAME-GRID passes (below): pass + 1 in the low bits, and
constexpr uint32_t ST_BORDER = 0x40000000u;      // a border pixel of the tile inside this quadrant changed
constexpr uint32_t ST_SELF = 0x80000000u;        // (word 0) the tile stopped at its round cap: it goes on itself
constexpr uint32_t ST_PASS = 0x3FFFFFFFu;
constexpr uint32_t RX_CAND = 64;      // candidates a workgroup collects before it hands them in (k_relax, append_flush)
// tile_list header: [0 .. 3] list lengths and [4 .. 7] entry tickets of pass & 3 (a launch clears the words of pass + 2)
constexpr uint32_t RL_HDR = 192;
// ... and, for the persistent tile-queue pass (k_relax, PERSIST): [8] tiles queued or running, [9] "the queue has run dry" (1) or
// "a worker ran out of its time budget" (2), [10] tile runs (diagnostics)
constexpr uint32_t RLQ_PHASE = 16;      // -DWS_TUNING: ticks of thread 0 per phase of a tile run, summed (load, rounds, epilogue, hand-in)
// Every word that all workers hammer sits on a 128-byte line of its own: head, tail, pending and the end flag in ONE line
// were ~100 atomics and polls per microsecond on one L2 channel, and an atomic round trip took 3 us (a tile run 49 us
// instead of 15).
constexpr uint32_t RLQ_HEAD = 32, RLQ_TAIL = 64, RLQ_PENDING = 96, RLQ_DONE = 128;
constexpr uint32_t RLQ_RUNS = 10, RLQ_WAIT = 11, RLQ_LIFE = 12, RLQ_POLLS = 13;      // (11-13: all workers' waiting / life time in 10 ns ticks, polls)
// A worker gives up -- and tells the others to -- when the launch has lasted this long (s_memrealtime ticks of 10 ns): no spin
// of this kernel can outlive it, whatever goes wrong with the queue.  What is left undone is work for the passes that follow.
constexpr unsigned long long RLQ_BUDGET_TICKS = 5000000ull;      // 50 ms; a smooth 8192^2 map needs 3
// Rounds per tile run of the persistent pass.  The ordinary late passes stop a tile after three (a pass ends when its slowest
// tile ends); without a pass barrier that reason is gone and a run's fixed costs (loads, stores, queue: ~10 us) are spread over
// more rounds: 8192^2 smooth maps, correlation 16 px: 6.74 ms with three, 6.36 with six, 6.28 with twelve.
constexpr uint32_t RLQ_ROUND_CAP = 6;
// PERSIST == 2, the queue in flood order: a worker takes a tile from the LOWEST non-empty of PQ_B buckets; a tile's bucket is the
// level (>> pq_shift) of the smallest stamp waiting at its borders.  tools/sim_tile_schedule.c (SIM_QUEUE=prio): on an 8192^2
// map of correlation length 64 px first-come order needs 152 k tile runs, this order 87 k with 32 buckets (86 k with 256):
// a tile that waits until the flood below it has passed runs once on final borders instead of once per arrival.
//   state   one word per tile: bit 31 running; bits 0 .. 30 "a stamp of bucket b waits" (idle: 0).  A tile that is not running and
//           has bits set is queued: its bit is set in the bitmap of its lowest bucket;
//   bucket  a bitmap over the tiles (idempotent: no ring, no overflow, no lap) and a count of its set bits; the 31 counts and a
//           copy of the end flag share ONE 128-byte line, so that a worker's look at all of them is one memory request (a line
//           per count: 32 requests per look, and the idle workers' looks alone slowed every tile load from 4 us to 21).
//           A stale bit (its tile runs, or has run from a lower bucket) costs a failed claim or one idle run.
constexpr int PQ_B = 31;
constexpr uint32_t PQ_RUNNING = 0x80000000u;
constexpr uint32_t PQ_HDR = 32;      // the counts' line: [b] set bits of bucket b, [31] the end flag again
__host__ __device__ inline size_t pq_base(uint32_t list_cap) { return (RL_HDR + 3 * (size_t)list_cap + 64 + 31) & ~(size_t)31; }
__host__ __device__ inline uint32_t pq_words_per_bucket(uint32_t tiles) { return ((tiles + 31u) / 32u + 255u) & ~255u; }      // whole 1 KiB chunks: one load of a wave
__host__ __device__ inline uint32_t pq_shift_of(uint32_t max_level) { return 24u + (max_level >= 124u ? 3u : max_level >= 62u ? 2u : max_level >= 31u ? 1u : 0u); }

// Stamp accesses of the persistent pass: tiles hand their border pixels to each other INSIDE a launch, across CUs and XCDs,
// so every stamp is stored write-through and loaded past L1 at agent scope (global_store / global_load ... sc1;
// MI355X_MICROARCH.md, inter-workgroup visibility).  Inline assembly, because HIP's agent-scope atomic loads are 8 bytes at
// most and each is waited for on its own: 24 dependent round trips per lane and tile run (52 us per run against 12).  The
// loads below are issued back to back and waited for ONCE (the wait's operands tie the loaded registers to it, so that no use
// can be scheduled in front of it).
__device__ __forceinline__ void coh_load4(u32x4_t &v, const uint32_t *p) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void coh_load1(uint32_t &v, const uint32_t *p) {
  asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
}
__device__ __forceinline__ void coh_store4(uint32_t *p, u32x4_t v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

template <int TW, int TH>
__device__ __forceinline__ unsigned long long relax_todo(int first, int stride, int chunk, int H, int W, int tilesX, int tilesY, int otherX,
                                                         int otherY, int shifted, uint32_t pass,
                                                         const uint32_t *__restrict__ stamps_prev, int read_same = 0) {
  const int lane = threadIdx.x & 63;
  const int ox = shifted ? TW / 2 : 0, oy = shifted ? TH / 2 : 0;
  const int t = first + lane * stride;
  const bool mine = lane < chunk && t < tilesX * tilesY;
  const int tx = mine ? t % tilesX : 0, ty = mine ? t / tilesX : 0;
  // a shifted grid can have a last row/column outside the plane
  bool run = mine && tx * TW - ox < W && ty * TH - oy < H;
  if (read_same) {
    // Same-grid passes (the long-range regime): the previous pass ran on THIS grid.  A tile runs when it stopped at its
    // round cap itself, or when a 4-neighbour changed a border pixel on the side facing it -- the two quadrants of the
    // neighbour that hold that side (quadrant = 2 * lower half + right half).  Exact for the same reason as the
    // alternating grids: an equation can only be left violated next to a border pixel that changed after it was read.
    const size_t tt = (size_t)ty * tilesX + tx;
    const uint32_t s0 = stamps_prev[tt * 4];
    bool flagged = (s0 & ST_PASS) == pass && (s0 & ST_SELF) != 0u;
    const int nx[4] = {tx, tx, tx - 1, tx + 1}, ny[4] = {ty - 1, ty + 1, ty, ty};
    const int qa[4] = {2, 0, 1, 0}, qb[4] = {3, 1, 3, 2};      // up: its bottom quadrants; down: top; left: right; right: left
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool ok = nx[k] >= 0 && nx[k] < tilesX && ny[k] >= 0 && ny[k] < tilesY;
      const size_t ot = (size_t)(ok ? ny[k] : 0) * tilesX + (ok ? nx[k] : 0);
      const uint32_t a = stamps_prev[ot * 4 + qa[k]], b = stamps_prev[ot * 4 + qb[k]];
      flagged |= ok && (((a & ST_PASS) == pass && (a & ST_BORDER) != 0u) || ((b & ST_PASS) == pass && (b & ST_BORDER) != 0u));
    }
    run = run && flagged;
  } else if (pass != 0) {
    // Quadrant (qx, qy) of a tile is quadrant (1-qx, 1-qy) of one tile of the previous pass's grid:
    // did that tile change a border pixel there?  Four independent loads on clamped indices.
    bool flagged = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int qx = q & 1, qy = q >> 1;
      const int oi = tx + qx - (shifted ? 1 : 0), oj = ty + qy - (shifted ? 1 : 0);
      const bool ok = oi >= 0 && oi < otherX && oj >= 0 && oj < otherY;
      const size_t ot = (size_t)(ok ? oj : 0) * otherX + (ok ? oi : 0);
      const uint32_t st = stamps_prev[ot * 4 + (3 - q)];
      flagged |= ok && st == pass;
    }
    run = run && flagged;
  }
  return __builtin_amdgcn_ballot_w64(run);
}

// CHUNKED = false: one tile per workgroup -- the passes in which (nearly) every tile runs.
// CHUNKED = true: `chunk` (<= 64) consecutive tiles per workgroup, run one after the other -- the late
// passes, in which few tiles run: a pass with nothing to do costs 1/chunk of the workgroup launches.
// (One body for both, not a shared device function: at the 80-VGPR cap the out-of-line form spilled.)
// ---- long-range rows ---------------------------------------------------------------------------
//
// A column sweep moves information one patch (4 pixels) per round along a row, so a flood that has to
// cross a 256-pixel tile sideways -- the normal case on a smooth map -- took 64 rounds per tile and pass.
// The row recurrence  t_x <- min(t_x, max(b_x, t_{x-1} + 1))  is a chain of clamped increments
// f_x(v) = med3(b_x, v + 1, t_x), and those compose:  (g o f)(v) = med3(lo, v + a, hi)  with
//   a = a_f + a_g,  lo = med3(lo_g, lo_f + a_g, hi_g),  hi = med3(lo_g, hi_f + a_g, hi_g).
// So the whole row is evaluated EXACTLY -- the same values the sequential sweep would give, pixel by
// pixel, ring carries included -- by a 6-step inclusive scan over the lanes of the wave: each lane
// reduces its 4 pixels to one (lo, hi, 4) triple, the scan composes triples, and every lane then knows
// the value that enters it from its neighbour.  Used only from pass RX_SCAN_FROM_PASS on (the bench field
// has converged by then; the scan costs registers, so the early passes run a variant without it).
// (r2: scans from the first round on and a cap on the rounds per tile run in those passes -- 8192^2 smooth maps, correlation
// length 16 / 64 / 256 px: 16.4 -> 13.1, 29.2 -> 22.2, 12.5 -> 10.7 ms with a cap of two (gpurun_out/r2e); three since the
// rounds of those passes are scans only and the scans DPP shifts: 8.4 -> 7.2, 8.6 -> 8.3, 4.1 -> 4.0 ms, gpurun_out/r2w/dpp.log)
constexpr uint32_t RX_SCAN_FROM_PASS = 4;
constexpr uint32_t RX_EARLY_ROUND_CAP = 4;     // rounds per tile run in passes 1 .. RX_SCAN_FROM_PASS - 1 (0: no cap): smooth 8192^2, correlation 16 px: 12.4 -> 10.4 ms
constexpr uint32_t RX_LATE_ROUND_CAP = 3;      // rounds per tile run from pass RX_SCAN_FROM_PASS on (0: no cap); see relax_pass

template <bool TRACK, bool RIGHT, int LX>
__device__ __forceinline__ void scan_row(uint32_t (&t)[RX_P], const uint32_t (&b)[RX_P], uint32_t halo_in, int xl, bool &changed) {
  // LX lanes make a tile row (64, or 32 when a wave holds two half-width bands: xl = lane & 31, and nothing crosses lane 32)
  // pixels in sweep order
  const uint32_t t0 = RIGHT ? t[0] : t[3], t1 = RIGHT ? t[1] : t[2], t2 = RIGHT ? t[2] : t[1], t3 = RIGHT ? t[3] : t[0];
  const uint32_t b0 = RIGHT ? b[0] : b[3], b1 = RIGHT ? b[1] : b[2], b2 = RIGHT ? b[2] : b[1], b3 = RIGHT ? b[3] : b[0];
  // this lane's four pixels as one function: its value for a huge and for a tiny argument
  uint32_t hi = t0;
  hi = med3u(b1, hi + 1u, t1); hi = med3u(b2, hi + 1u, t2); hi = med3u(b3, hi + 1u, t3);
  uint32_t lo = b0;
  lo = med3u(b1, lo + 1u, t1); lo = med3u(b2, lo + 1u, t2); lo = med3u(b3, lo + 1u, t3);
  // Every lane adds the same RX_P to what passes through it, so in the coordinate W = v + RX_P * (lanes still to go, this
  // one included) a lane is a PURE clamp [lo + bias, hi + bias], and clamps compose by two medians -- no running sum to
  // carry through the scan, no additions inside it (r2: a third fewer instructions per step, two shuffles instead of three)
  const uint32_t bias = RIGHT ? (uint32_t)(RX_P * (LX - 1 - xl)) : (uint32_t)(RX_P * xl);
  lo += bias;
  hi += bias;
  // Inclusive scan in sweep order, acc_i <- acc_(i -/+ o) then acc_i, WITHOUT LDS round trips (__shfl_up is a ds_bpermute:
  // ~100 cycles a step, twelve dependent steps per row and direction, in passes that run at 40 % VALU use): inside a row
  // of 16 lanes four DPP row shifts whose `old` operand is the identity for lanes with nothing before them (0 for the
  // lower bound: med3(lo, 0, hi) = lo; ~0 for the upper one), then the rows' totals across rows -- DPP row broadcasts
  // going right, v_readlane of the rows' first lanes going left (GFX9 has no broadcast in that direction).
  auto combine = [&](uint32_t plo, uint32_t phi) {
    const uint32_t nlo = med3u(lo, plo, hi), nhi = med3u(lo, phi, hi);      // the earlier clamp, then this one
    lo = nlo; hi = nhi;
  };
#define WS_DPP(old, v, ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp((int)(old), (int)(v), ctrl, rows, 0xF, false)
  if (RIGHT) {
    combine(WS_DPP(0u, lo, 0x111, 0xF), WS_DPP(~0u, hi, 0x111, 0xF));      // row_shr:1
    combine(WS_DPP(0u, lo, 0x112, 0xF), WS_DPP(~0u, hi, 0x112, 0xF));      // row_shr:2
    combine(WS_DPP(0u, lo, 0x114, 0xF), WS_DPP(~0u, hi, 0x114, 0xF));      // row_shr:4
    combine(WS_DPP(0u, lo, 0x118, 0xF), WS_DPP(~0u, hi, 0x118, 0xF));      // row_shr:8
    combine(WS_DPP(0u, lo, 0x142, 0xA), WS_DPP(~0u, hi, 0x142, 0xA));      // row_bcast:15 into rows 1 and 3
    if (LX == 64) combine(WS_DPP(0u, lo, 0x143, 0xC), WS_DPP(~0u, hi, 0x143, 0xC));      // row_bcast:31 into rows 2 and 3
  } else {
    combine(WS_DPP(0u, lo, 0x101, 0xF), WS_DPP(~0u, hi, 0x101, 0xF));      // row_shl:1
    combine(WS_DPP(0u, lo, 0x102, 0xF), WS_DPP(~0u, hi, 0x102, 0xF));      // row_shl:2
    combine(WS_DPP(0u, lo, 0x104, 0xF), WS_DPP(~0u, hi, 0x104, 0xF));      // row_shl:4
    combine(WS_DPP(0u, lo, 0x108, 0xF), WS_DPP(~0u, hi, 0x108, 0xF));      // row_shl:8
    {
      const int lane = (int)(threadIdx.x & 63u);
      const uint32_t l16 = (uint32_t)__builtin_amdgcn_readlane((int)lo, 16), h16 = (uint32_t)__builtin_amdgcn_readlane((int)hi, 16);
      const uint32_t l48 = (uint32_t)__builtin_amdgcn_readlane((int)lo, 48), h48 = (uint32_t)__builtin_amdgcn_readlane((int)hi, 48);
      const bool r0 = lane < 16, r2 = lane >= 32 && lane < 48;      // rows 0 and 2 take the total of the row after them
      combine(r0 ? l16 : (r2 ? l48 : 0u), r0 ? h16 : (r2 ? h48 : ~0u));
      if (LX == 64) {
        const uint32_t l32 = (uint32_t)__builtin_amdgcn_readlane((int)lo, 32), h32 = (uint32_t)__builtin_amdgcn_readlane((int)hi, 32);
        combine(lane < 32 ? l32 : 0u, lane < 32 ? h32 : ~0u);
      }
    }
  }
#undef WS_DPP
  // the value that leaves this lane when `halo_in` enters the row, handed to the next lane
  const uint32_t leaves = med3u(lo, halo_in + (uint32_t)(RX_P * LX), hi) - bias;
  uint32_t vin = RIGHT ? lane_left(halo_in, leaves) : lane_right(halo_in, leaves);
  if (LX != 64) vin = (RIGHT ? xl == 0 : xl == LX - 1) ? halo_in : vin;      // the first lane of the second half-row
  const uint32_t n0 = med3u(b0, vin + 1u, t0), n1 = med3u(b1, n0 + 1u, t1), n2 = med3u(b2, n1 + 1u, t2), n3 = med3u(b3, n2 + 1u, t3);
  if (TRACK) changed |= (n0 != t0) | (n1 != t1) | (n2 != t2) | (n3 != t3);
  t[RIGHT ? 0 : 3] = n0; t[RIGHT ? 1 : 2] = n1; t[RIGHT ? 2 : 1] = n2; t[RIGHT ? 3 : 0] = n3;
}

// The same idea down AND up the columns of a tile, in one phase: the rows of a column live in NB different bands (waves, or
// halves of waves), so every band publishes its four rows of each column as two clamped increments in LDS -- entered from
// above, entered from below; two waves then walk the columns once -- down and up -- through those functions from the tile's
// halo rows and leave every band the value that enters it; then every band takes its own rows downwards from what came
// from above, and upwards from what came from below.  Exact relaxation steps, like the row scan; what enters from the other
// bands is their state when the phase began.
// (r2, first form: a scan down and a scan up, each with its own barriers and with 0 .. NB - 1 dependent reads depending
// on the band -- the last band walked 15 bands while the others waited, twice per round: 4 of the 6.6 us of a tile run's
// rounds, tools/diag_relax_smooth.hip.  Second form: every band composed the NB - 1 others by itself, one barrier.
// tools/sim_tile_schedule.c, recipes "rdlu|L|" and "rlc|L|": same passes, tile runs and rounds either way.)
template <int NB, int TW, bool SPLIT>
__device__ __forceinline__ void scan_cols_both(patch_t &T, const patch_t &B, uint32_t *fn, const uint32_t *halo_top, const uint32_t *halo_bottom,
                                               int band, int xl) {      // fn: [2][NB][2][TW]: direction, band, (lo, hi)
  auto slot = [&](int dir, int k, int which) { return fn + (((size_t)dir * NB + k) * 2 + which) * TW + xl * RX_P; };
  {
    uint32_t lo[RX_P], hi[RX_P];
#pragma unroll
    for (int dir = 0; dir < 2; ++dir) {      // 0: entered from above, rows 0 .. 3; 1: from below, rows 3 .. 0
#pragma unroll
      for (int c = 0; c < RX_P; ++c) {
        uint32_t h = T[dir ? 3 : 0][c], l = B[dir ? 3 : 0][c];
#pragma unroll
        for (int k = 1; k < RX_P; ++k) {
          const int r = dir ? RX_P - 1 - k : k;
          h = med3u(B[r][c], h + 1u, T[r][c]);
          l = med3u(B[r][c], l + 1u, T[r][c]);
        }
        lo[c] = l; hi[c] = h;
      }
      *reinterpret_cast<u32x4_t *>(slot(dir, band, 0)) = u32x4_t{lo[0], lo[1], lo[2], lo[3]};
      *reinterpret_cast<u32x4_t *>(slot(dir, band, 1)) = u32x4_t{hi[0], hi[1], hi[2], hi[3]};
    }
  }
  __syncthreads();
  // What enters each band from above and from below is a chain through the bands' functions -- NB - 1 steps, but ONE chain
  // per column and direction, not one per band: wave 0 walks down, wave 1 walks up, a lane per four columns, and leaves in
  // the (lo) slot of every band the value that enters it.  (Every band composing the bands before it by itself: NB - 1 LDS
  // reads of 32 bytes in every lane -- 1 MB per round and tile at sixteen waves, 3 of a round's 9 us.)
  constexpr int LXc = TW / RX_P;      // lanes across a tile row
  {
    const int w = (int)(threadIdx.x >> 6), l = (int)(threadIdx.x & 63);
    if (w < 2 && l < LXc) {
      const int dir = w;
      const u32x4_t h4 = *reinterpret_cast<const u32x4_t *>(&(dir ? halo_bottom : halo_top)[l * RX_P]);
      uint32_t v0 = h4.x, v1 = h4.y, v2 = h4.z, v3 = h4.w;
#pragma unroll 4
      for (int i = 0; i < NB; ++i) {
        const int k = dir ? NB - 1 - i : i;
        uint32_t *lo_p = fn + (((size_t)dir * NB + k) * 2 + 0) * TW + l * RX_P, *hi_p = lo_p + TW;
        const u32x4_t l4 = *reinterpret_cast<const u32x4_t *>(lo_p);
        const u32x4_t g4 = *reinterpret_cast<const u32x4_t *>(hi_p);
        *reinterpret_cast<u32x4_t *>(lo_p) = u32x4_t{v0, v1, v2, v3};      // what enters band k
        v0 = med3u(l4.x, v0 + RX_P, g4.x); v1 = med3u(l4.y, v1 + RX_P, g4.y);
        v2 = med3u(l4.z, v2 + RX_P, g4.z); v3 = med3u(l4.w, v3 + RX_P, g4.w);
      }
    }
  }
  __syncthreads();
  const u32x4_t d4 = *reinterpret_cast<const u32x4_t *>(slot(0, band, 0));
  const u32x4_t u4 = *reinterpret_cast<const u32x4_t *>(slot(1, band, 0));
  const uint32_t vd[RX_P] = {d4.x, d4.y, d4.z, d4.w}, vu[RX_P] = {u4.x, u4.y, u4.z, u4.w};
#pragma unroll
  for (int c = 0; c < RX_P; ++c) {
    uint32_t n = vd[c];
#pragma unroll
    for (int r = 0; r < RX_P; ++r) { n = med3u(B[r][c], n + 1u, T[r][c]); T[r][c] = n; }
    n = vu[c];
#pragma unroll
    for (int r = RX_P - 1; r >= 0; --r) { n = med3u(B[r][c], n + 1u, T[r][c]); T[r][c] = n; }
  }
  // (no barrier here: two lie between this phase and the next write of the functions -- the end of the round's free part
  // and the checked sweep's)
}

// PERSIST (the first same-grid pass of a long-range flood, relax_pass): ONE launch instead of a pass per step of the flood
// front.  The workgroups -- all resident -- pull tiles from a queue; a tile run that changes a side that matters to a
// neighbour, or stops at its round cap, puts that neighbour (itself) back into the queue, at once: no pass barrier, so the
// critical path is the chain of tile runs along the flood instead of the slowest tile of each of ~180 launches.
//   queue   a ring of (sequence number, tile) pairs in the two entry arrays of tile_list; head = the ticket word, tail = the
//           length word of this pass; a worker draws a ticket and waits for ITS entry (the sequence number tells it from the
//           ring's previous lap);
//   state   one word per tile (the "queued" marks): bit 0 queued -- or, while bit 1 is set, "flagged again while running" --,
//           bit 1 running.  A tile is in the queue at most once, and a tile flagged while it runs queues itself when it ends;
//   end     [RLQ_PENDING] counts tiles queued or running; who brings it to zero raises [RLQ_DONE].
// Exactness does not rest on any of this: stamps only ever fall, by relaxation steps from upper bounds (a stale read is an
// older, larger stamp: less progress, never a wrong value), and the pass AFTER this launch runs every tile once from an
// all-tiles list -- the flood is at its fixpoint when the ordinary passes that follow say so.
template <int NW, bool CHUNKED, bool SCAN, bool LITE, bool SPLIT = false, int SEAM = 0, int PERSIST = 0>
__global__ __launch_bounds__(64 * NW, SCAN ? 4 : 6) void k_relax(      // the scan variant trades occupancy (few tiles run there) for registers
const uint8_t *__restrict__ img, size_t img_stride, uint32_t *keys,
                                                      int H, int W, int tilesX, int tilesY, int otherX, int otherY,
                                                      int shifted, int chunk, uint32_t max_level, uint32_t pass,
                                                      const uint32_t *__restrict__ stamps_prev, uint32_t *stamps_cur,
                                                      PassFlags pf, uint32_t max_iters,
                                                      const uint32_t *__restrict__ seed_labels, int seed_bits, int SH, int check_carry,
                                                      int pad, uint32_t *tile_list, int use_list, int read_same,
                                                      int write_same, uint32_t list_cap, int append_next) {
  // SH: rows per slice.  A batch of independent slices is one plane of H = S * SH rows in which the first and last row
  // of every slice are image-border rows (never flooded: walls between the slices); SH == H for a single image.
  // SPLIT (the same-grid passes of a long-range flood): a wave is TWO bands of half the width -- lanes 0..31 hold four rows
  // of 128 columns, lanes 32..63 the four rows below them -- so the tile is 128 x 64 instead of 256 x 32.  On a smooth map
  // the number of passes is set by how far a pass carries a flood vertically, and a pass costs its tile runs whatever
  // their shape: tools/sim_tile_schedule.c has a quarter to a third fewer of both for the squarer tile.
  constexpr int LX = SPLIT ? 32 : 64;             // lanes across a tile row
  constexpr int NB = SPLIT ? 2 * NW : NW;         // bands of RX_P rows
  constexpr int TW = LX * RX_P, TH = NB * RX_P;
  // SEAM (relax_pass, seam repair: the pass after pass 0 of a transform that starts from its seeds).  What a pass leaves
  // wrong lies within a few pixels of its tile borders, so the pass after pass 0 only has to look along those:
  //   SEAM 1  a tile is 256 x 8 pixels (NW = 2) astride a horizontal seam of the 256 x 32 grid, rows 32 (ty + 1) - 4 ...;
  //   SEAM 2  a tile is a 32-row slice of 32 vertical seams: lanes 2j and 2j + 1 hold the 8 columns astride seam
  //           32 tile_x + j + 1 (x = 256 times that), and no stamp crosses from one lane PAIR to the next.
  // A seam tile iterates to its fixpoint and raises, in the stamp word that the next pass of the anchored grid reads, the
  // flag of every 256 x 32 tile that holds a pixel next to a changed outer row or column of it.  (otherX: that array's pitch.)
  constexpr int SEAM_PY = 32, SEAM_PX = 256, SEAM_HALF = SEAM == 1 ? TH / 2 : 4;
  // row 0: halo above the tile; rows 1+2w / 2+2w: top / bottom row of band w; last row: halo below
  __shared__ __attribute__((aligned(16))) uint32_t sRow[2 * NB + 2][TW];
  // the tile border as loaded (top row, bottom row, left column, right column): compared with the
  // final values to tell which tile edges changed; parked in LDS to keep the VGPR count at 80
  __shared__ __attribute__((aligned(16))) uint32_t sInitRow[2][TW];
  __shared__ uint32_t sInitCol[2][TH];
  __shared__ uint32_t s_edges;
  // list mode: the tiles this workgroup's runs want in the next pass's list, handed in together -- a "queued" exchange
  // and a list ticket are two dependent atomic round trips, 3 of the 4.5 us a tile run's epilogue took with seven of the
  // eight waves idle (tools/diag_relax_smooth.hip); a workgroup runs ~5 tiles in a heavy pass and now pays them once
  __shared__ uint32_t s_cand[RX_CAND];
  __shared__ uint32_t s_ncand;
  __shared__ uint32_t s_next[2];             // list mode: the entry this workgroup takes after the current one (by run parity)
  // "some lane changed in iteration k" lives in slot k % 3: written before barrier k, read after it,
  // cleared by thread 0 for iteration k + 2 -- one barrier per iteration instead of the three a
  // __syncthreads_or costs
  __shared__ uint32_t s_flag[3];
  __shared__ uint64_t s_sum[64 * NW];        // per-lane patch checksum taken at load time (parked: VGPRs are at the cap)
  // long-range columns (SCAN): every band's four rows of a column as one clamped increment (lo, hi)
  __shared__ __attribute__((aligned(16))) uint32_t sFn[2][SCAN ? NB : 1][2][SCAN ? TW : 4];


  // the first wave of workgroup 0 clears the next pass's convergence slot
  if (blockIdx.x == 0 && threadIdx.x < NSTRIPE)
    pf.edge_changed[((pass + 1) % COUNTER_RING) * FLAG_SLOT + threadIdx.x * STRIPE_STRIDE] = 0;
  // ... and the length of the next pass's tile list (k_relax_list appends to it after this launch has ended)
  // (tile_list: [0..3] list lengths of pass & 3; [4 ...) entries of the even passes, then of the odd ones, then one
  // "queued for pass" word per tile.  This launch reads the list of `pass`, may append to the list of pass + 1, and clears
  // the length that pass + 2 will count up from.)
  if (tile_list && blockIdx.x == 0 && threadIdx.x == 0) { tile_list[(pass + 2) & 3u] = 0u; tile_list[4 + ((pass + 2) & 3u)] = 0u; }
  // (XCD-aware: consecutive workgroups go to different XCDs; see xcd_span_index)
  // A chunk is `chunk` tiles one grid size apart, not neighbours: on a smooth map the tiles that still
  // run line up along a front, and four neighbours in one workgroup ran one after the other.
  const int first = (int)xcd_span_index(blockIdx.x, gridDim.x);
  const int stride = (int)gridDim.x;
  unsigned long long todo = 1;
  // list mode (late passes of a long-range flood): the tiles to run were compacted by k_relax_list; workgroup b takes
  // entries b, b + gridDim.x, ... -- no workgroup is launched for a tile that has nothing to do, none owns two busy ones
  uint32_t entry = blockIdx.x, n_entries = 0, first_entry = 0, runs_done = 0;
  // PERSIST: the queue (see above)
  unsigned long long *q_ring = reinterpret_cast<unsigned long long *>(tile_list + RL_HDR);      // list_cap entries: (sequence + 1) << 32 | tile
  uint32_t *q_state = tile_list + RL_HDR + 2 * (size_t)list_cap;
  uint32_t *q_head = tile_list + RLQ_HEAD, *q_tail = tile_list + RLQ_TAIL;
  uint32_t *q_pending = tile_list + RLQ_PENDING, *q_done = tile_list + RLQ_DONE;
  __shared__ uint32_t s_qtile, s_qbucket, s_handoff;
  __shared__ uint32_t s_sidemin[4];      // PERSIST == 2: the smallest new stamp that matters across the top / bottom / left / right border
  uint32_t *pq_avail = tile_list + pq_base(list_cap);
  const uint32_t pq_bw = pq_words_per_bucket((uint32_t)(tilesX * tilesY));
  uint32_t *pq_bits = pq_avail + PQ_HDR;
  const uint32_t pq_shift = pq_shift_of(max_level);
  unsigned long long q_t0 = 0;
  uint32_t q_wait = 0, q_polls = 0;
#ifdef WS_TUNING
  uint32_t q_ph[4] = {0, 0, 0, 0};
  unsigned long long q_tp = 0;
#define WS_QPHASE(k) do { if (PERSIST && threadIdx.x == 0) { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); q_ph[k] += (uint32_t)(n_ - q_tp); q_tp = n_; } } while (0)
#define WS_QPHASE0 do { if (PERSIST && threadIdx.x == 0) q_tp = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define WS_QPHASE(k) do {} while (0)
#define WS_QPHASE0 do {} while (0)
#endif
#ifdef WS_TUNING
  const unsigned long long q_c0 = PERSIST ? __builtin_amdgcn_s_memtime() : 0ull;
#endif
  if (PERSIST) {
    q_t0 = __builtin_amdgcn_s_memrealtime();
    // whatever this launch does, the passes after it look at every tile again: tell the host that they have to run
    if (blockIdx.x == 0 && threadIdx.x == 0 && __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
      pf.edge_changed[(pass % COUNTER_RING) * FLAG_SLOT] = 1u;
  } else if (CHUNKED && use_list) {
    // (the workgroup's first entry is asked for together with the list length, not after it: one memory round trip less
    // at the head of every tile run of a thin pass, which IS such a pass's length)
    first_entry = tile_list[RL_HDR + (pass & 1u) * list_cap + min(entry, list_cap - 1u)];
    n_entries = tile_list[pass & 3u];
    if (entry >= n_entries) return;
  } else if (CHUNKED && !PERSIST) {
    todo = relax_todo<TW, TH>(first, stride, chunk, H, W, tilesX, tilesY, otherX, otherY, shifted, pass, stamps_prev, read_same);
    if (todo == 0) return;
  } else if (!PERSIST) {
    const int tx = first % tilesX, ty = first / tilesX;
    // (a seam whose tiles on both sides have asked for a re-run already -- pass 0 stopped at its round cap there: a smooth
    // map -- is left to them: on such maps the repair would be 64 us of sweeps that the re-runs undo)
    if (SEAM == 1) {
      if (tx * TW >= W || (ty + 1) * SEAM_PY >= H) return;
      // (a tile that asked for its re-run is work for pass 2: this pass must not look like a fixpoint to the host -- pass 0
      // cannot say so itself, its launch clears this pass's convergence slot)
      const bool fa = stamps_cur[((size_t)ty * otherX + tx + 1) * 4 + 2] == pass + 1, fb = stamps_cur[((size_t)(ty + 1) * otherX + tx + 1) * 4 + 2] == pass + 1;
      if ((fa || fb) && threadIdx.x == 0) pf.edge_changed[(pass % COUNTER_RING) * FLAG_SLOT + (blockIdx.x % NSTRIPE) * STRIPE_STRIDE] = 1u;
      if (fa && fb) return;
    } else if (SEAM == 2) {
      if (ty * TH - (shifted ? TH / 2 : 0) >= H || (tx * 32 + 1) * SEAM_PX >= W) return;
      const int l = threadIdx.x & 63;
      const int sxl = (tx * 32 + (l >> 1) + 1) * SEAM_PX;
      const bool settled = sxl >= W || stamps_cur[((size_t)((ty * TH) / SEAM_PY) * otherX + (sxl / SEAM_PX - 1 + (l & 1)) + 1) * 4 + 2] == pass + 1;
      if (__builtin_amdgcn_ballot_w64(settled) == ~0ull) return;
    }
    else if (tx * TW - (shifted ? TW / 2 : 0) >= W || ty * TH - (shifted ? TH / 2 : 0) >= H) return;
    if (SEAM == 0 && pass != 0) {       // the same test as relax_todo, on scalars (workgroup uniform)
      bool run = false;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int qx = q & 1, qy = q >> 1;
        const int oi = tx + qx - (shifted ? 1 : 0), oj = ty + qy - (shifted ? 1 : 0);
        const bool ok = oi >= 0 && oi < otherX && oj >= 0 && oj < otherY;
        const size_t ot = (size_t)(ok ? oj : 0) * otherX + (ok ? oi : 0);
        const uint32_t st = stamps_prev[ot * 4 + (3 - q)];
        run |= ok && st == pass;
      }
      if (!run) return;
    }
  }
  if (threadIdx.x == 0) { s_ncand = 0; if (PERSIST == 2) s_handoff = 0; }      // (read again only after the first tile's barriers)
  // Wave 0, all lanes: one "queued for pass p" exchange per candidate (a tile enters a list once: whoever finds the old
  // mark adds it), one ticket for the new entries of all of them.
  auto append_flush = [&]() {
    const int lane = threadIdx.x & 63;
    const uint32_t n = s_ncand;
    uint32_t *queued = tile_list + RL_HDR + 2 * (size_t)list_cap;
    uint32_t *next = tile_list + RL_HDR + ((pass + 1) & 1u) * (size_t)list_cap;
    const uint32_t mark = pass + 1;
    const bool mine = (uint32_t)lane < n;
    const uint32_t cand = mine ? s_cand[lane] : 0u;
    const bool fresh = mine && atomicExch(&queued[cand], mark) != mark;
    const unsigned long long fm = __builtin_amdgcn_ballot_w64(fresh);
    if (fm) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&tile_list[(pass + 1) & 3u], (uint32_t)__popcll(fm));
      at = __shfl(at, 0, 64);
      if (fresh) next[at + __popcll(fm & ((1ull << lane) - 1ull))] = cand;
    }
    if (lane == 0) s_ncand = 0;
  };
  for (;;) {
  uint32_t q_tile = 0;
  if (PERSIST == 2) {
    if (threadIdx.x < 64 && s_handoff != 0u) {      // (wave uniform) the run before this one took a tile it had announced itself
      if (threadIdx.x == 0) { s_handoff = 0u; s_sidemin[0] = s_sidemin[1] = s_sidemin[2] = s_sidemin[3] = 0xFFFFFFFFu; }
    } else if (threadIdx.x < 64) {
      const int ln = (int)threadIdx.x;
      const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
      const uint32_t nchunk = pq_bw >> 8;
      uint32_t got = 0, got_b = 0, idle = 0;
      for (;;) {
        // one look at every bucket's count (lane b) and at the end flag (lane 63)
        uint32_t av = 0;
        if (ln < 32) av = __hip_atomic_load(pq_avail + ln, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__shfl((int)av, PQ_B, 64) != 0) break;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ln < PQ_B && (int)av > 0);
        // (every turn of this loop checks the clock: whatever goes wrong with counts or bits, the end flag is seen a turn later)
        if (ln == 0 && __builtin_amdgcn_s_memrealtime() - q_t0 > RLQ_BUDGET_TICKS) {
          __hip_atomic_store(q_done, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(pq_avail + PQ_B, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // A look that ends without a tile -- nothing queued, or somebody else was quicker -- is followed by a pause that grows
        // with the looks in a row: a thousand workers after the same few bits, each look nine memory requests to the same
        // nine lines, held every load of the RUNNING tiles up behind them (a tile run 60 us instead of 15).  Work that
        // appears is found by whoever looks next, so the delay is the pause divided by the number of idle workers.
        auto pause = [&]() {
          ++q_polls;
          ++idle;
          // ... and with the worker's number: sixteen look every 3 us, forty-eight every 14, the rest every 54 -- a front that
          // is a chain of tile runs is followed by the worker that runs it (the hand-off at the end of a run), and a backlog
          // that lasts is found by everybody within one long pause
          const int reps = (use_list & 2) ? 1 : (blockIdx.x < 16u ? 1 : (blockIdx.x < 64u ? 4 : 16));
          if (use_list & 4) __builtin_amdgcn_s_sleep(2);
          else if (idle < 3u) __builtin_amdgcn_s_sleep(8);
          else if (idle < 6u) __builtin_amdgcn_s_sleep(64);
          else { for (int z = 0; z < reps; ++z) __builtin_amdgcn_s_sleep(127); }
        };
        if (m == 0) { pause(); continue; }
        const uint32_t b = (uint32_t)__builtin_ctzll(m);
        uint32_t *bm = pq_bits + (size_t)b * pq_bw;
        for (uint32_t cc = 0; cc < nchunk; ++cc) {
          const uint32_t c = (cc + blockIdx.x) % nchunk;      // (workers start in different chunks of a long bitmap)
          u32x4_t v;
          coh_load4(v, bm + c * 256u + (uint32_t)ln * 4u);
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(v) : : "memory");
          unsigned long long mm = __builtin_amdgcn_ballot_w64((v.x | v.y | v.z | v.w) != 0u);
          if (mm == 0) continue;
          // workers that look at the same moment take different bits: the k-th lane that has any
          int k = (int)(blockIdx.x % (uint32_t)__popcll(mm));
          while (k-- > 0) mm &= mm - 1ull;
          const int sel = (int)__builtin_ctzll(mm);
          uint32_t t1 = 0, tb = 0;
          if (ln == sel) {
            const int j = v.x ? 0 : (v.y ? 1 : (v.z ? 2 : 3));
            const uint32_t wv = j == 0 ? v.x : (j == 1 ? v.y : (j == 2 ? v.z : v.w));
            const uint32_t bit = wv & (0u - wv);
            const uint32_t wi = c * 256u + (uint32_t)ln * 4u + (uint32_t)j;
            if (atomicAnd(bm + wi, ~bit) & bit) {      // the bit is mine
              atomicSub(pq_avail + b, 1u);
              const uint32_t t = (wi << 5) + (uint32_t)__builtin_ctz(bit);
              // not running -> running, every waiting bit taken with it: what they announced was stored before they were set,
              // and this run loads after this exchange.  Running already (a stale bit): that run's end looks at the bits.
              const uint32_t so = atomicMax(&q_state[t], PQ_RUNNING);
              if (!(so & PQ_RUNNING)) {
                if (so == 0u) atomicAdd(q_pending, 1u);      // (a stale bit of an idle tile: it runs once for nothing)
                t1 = t + 1u;
                tb = so ? (uint32_t)__builtin_ctz(so) : b;
              }
            }
          }
          got = (uint32_t)__shfl((int)t1, sel, 64);
          got_b = (uint32_t)__shfl((int)tb, sel, 64);
          break;      // taken, or somebody else was quicker: look at the counts again
        }
        if (got) break;
        pause();
      }
      if (ln == 0) {
        q_wait += (uint32_t)(__builtin_amdgcn_s_memrealtime() - w0);
        s_qtile = got;
        s_qbucket = got_b;
        s_sidemin[0] = s_sidemin[1] = s_sidemin[2] = s_sidemin[3] = 0xFFFFFFFFu;
      }
    }
    __syncthreads();
    q_tile = s_qtile;
  } else if (PERSIST) {
    if (threadIdx.x == 0) {
      unsigned long long v = 0;
      if (__hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        const uint32_t my = atomicAdd(q_head, 1u);
        unsigned long long *slot = q_ring + (my % list_cap);
        for (;;) {
          v = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((uint32_t)(v >> 32) == my + 1u) break;      // my entry (not one of the ring's previous lap)
          v = 0;
          ++q_polls;
          if (__hip_atomic_load(q_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (__builtin_amdgcn_s_memrealtime() - q_t0 > RLQ_BUDGET_TICKS) { __hip_atomic_store(q_done, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
          // an idle worker must not cost the busy ones their memory bandwidth: the first polls come quickly (work usually
          // arrives within a tile run), later ones every few microseconds
          // Who is next in line polls quickly (the flood is often a chain of tile runs: this wait is on its critical path);
          // who is far behind the tail sleeps longer -- an idle worker must not cost the busy ones their memory bandwidth.
          const uint32_t behind = my - __hip_atomic_load(q_tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // tickets drawn before mine and not yet filled
          if (behind < 4u) __builtin_amdgcn_s_sleep(2);
          else if (behind < 32u) __builtin_amdgcn_s_sleep(24);
          else __builtin_amdgcn_s_sleep(127);
        }
        q_wait += (uint32_t)(__builtin_amdgcn_s_memrealtime() - w0);
        if (v != 0) atomicExch(q_state + (uint32_t)v, 2u);      // queued -> running: whoever flags it from now on makes it run again
      }
      s_qtile = v != 0 ? (uint32_t)v + 1u : 0u;
    }
    __syncthreads();
    q_tile = s_qtile;
  }
  if (PERSIST) {
    if (q_tile == 0) {           // workgroup uniform: the queue has run dry (or the time budget is spent)
      if (threadIdx.x == 0) {
        atomicAdd(tile_list + RLQ_WAIT, q_wait);
        atomicAdd(tile_list + RLQ_LIFE, (uint32_t)(__builtin_amdgcn_s_memrealtime() - q_t0));
        atomicAdd(tile_list + RLQ_POLLS, q_polls);
#ifdef WS_TUNING
        for (int k = 0; k < 4; ++k) atomicAdd(tile_list + RLQ_PHASE + k, q_ph[k]);
        atomicAdd(tile_list + RLQ_PHASE + 4, (uint32_t)((__builtin_amdgcn_s_memtime() - q_c0) >> 8));      // shader cycles / 256
#endif
      }
      break;
    }
  }
  // re-derived per tile on purpose (the asm hides the value from loop-invariant hoisting): hoisted
  // per-lane addresses pushed the chunked variant over the 80-VGPR cap and into scratch
  int tid = threadIdx.x;
  if (CHUNKED) asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int xl = SPLIT ? lane & 31 : lane;                                      // column block of the tile row
  const int band = SPLIT ? (tid >> 6) * 2 + (lane >> 5) : tid >> 6;             // four-row band of the tile
  const int tile = PERSIST ? (int)(q_tile - 1u)
                           : (CHUNKED ? (use_list ? (int)(entry == blockIdx.x ? first_entry : tile_list[RL_HDR + (pass & 1u) * list_cap + entry]) : first + (int)__builtin_ctzll(todo) * stride) : first);
  const int tile_x = tile % tilesX, tile_y = tile / tilesX;
  const int x0 = SEAM ? tile_x * TW : tile_x * TW - (shifted ? TW / 2 : 0);
  const int y0 = SEAM == 1 ? (tile_y + 1) * SEAM_PY - SEAM_HALF : tile_y * TH - (shifted ? TH / 2 : 0);
  const int seam_x = (tile_x * 32 + (lane >> 1) + 1) * SEAM_PX;      // SEAM 2: this lane pair's seam (outside the plane: no seam)

  WS_STAMP(0);
  WS_QPHASE0;
  const int gx0 = SEAM == 2 ? (seam_x < W ? seam_x - SEAM_HALF + (lane & 1) * RX_P : W) : x0 + xl * RX_P, gyb = y0 + band * RX_P;
  if (tid == 0) { s_edges = 0; s_flag[0] = 0; s_flag[1] = 0; s_flag[2] = 0; }
  // List mode: entries are handed out by ticket, not in strides of the grid -- tile runs last 8 to 20 us, and with a fixed
  // share a pass ended with most workgroups gone and a few still on their third tile.  The ticket for the NEXT entry is
  // drawn now and read after the run: its round trip hides behind the tile.
  if (!PERSIST && CHUNKED && use_list && tid == 0) s_next[runs_done & 1u] = gridDim.x + atomicAdd(&tile_list[4 + (pass & 3u)], 1u);

  // ---- load phase ---------------------------------------------------------------------------
  uint32_t T[RX_P][RX_P], B[RX_P][RX_P], halo[RX_P];
  // fast path (workgroup uniform): the tile lies inside the image in x and image rows can be read
  // as aligned dwords -> one 16-byte stamp load and one 4-byte image load per lane and row
  // (patches outside the plane read a clamped address and are masked afterwards: with W % 4 == 0 a
  // patch is either wholly inside or wholly outside)
  // (pad: the image is the caller's unpadded one, read through padded_img_index -- byte loads)
  const bool fast = !pad && W >= RX_P && ((SEAM != 2 && x0 >= 0 && x0 + TW <= W) || (W & 3) == 0) &&
                    ((reinterpret_cast<uintptr_t>(img) | img_stride) & 3u) == 0 && img_stride <= 0xFFFFFFFFull;
  // Index arithmetic of the fast paths in 32 bits where the plane allows it: the first form of these loads spent 22 vector
  // instructions per patch row on addresses, five of them 64-bit multiplies (v_mad_i64_i32, v_mul_lo_u32: quarter rate).
  const uint32_t stride32 = (uint32_t)img_stride;
  const bool dims24 = (uint32_t)W < (1u << 24) && (uint32_t)H < (1u << 24);      // kernel uniform: row * W as v_mul_u32_u24
  const int gxc0 = min(max(gx0, 0), max(W - RX_P, 0));
  // tile halo columns: the left half of a row's lanes fetch the column left of the tile, the right half the one right
  // of it; only the first / last lane of a row ever use the value (as the DPP `old` operand)
  // (SEAM 2: every lane is the first or the last of its pair's row)
  const int xh_raw = SEAM == 2 ? ((lane & 1) ? gx0 + RX_P : gx0 - 1) : (xl < LX / 2 ? x0 - 1 : x0 + TW);
  const int xh = min(max(xh_raw, 0), W - 1);
  const bool xh_ok = SEAM == 2 ? (xh_raw >= 0 && xh_raw < W && seam_x < W) : (xl < LX / 2 ? x0 > 0 : x0 + TW < W);
  const int gy_halo_raw = band == 0 ? y0 - 1 : y0 + TH;
  const int gy_halo = min(max(gy_halo_raw, 0), H - 1);
  u32x4_t halo_row;
  // pass 0 of a whole-image transform reads the freshly painted LABEL plane instead of a stamp
  // plane (seed = coloured pixel = stamp 0, everything else never-coloured): nobody has to fill the
  // stamp plane first, this pass writes all of it
  // (seed_bits: the "plane" is one bit per pixel instead -- the side table of a strictly increasing
  // seed list, ws_kernels.hip; only offered for W % 4 == 0, so a patch row is one nibble of one word)
  const bool from_labels = seed_labels != nullptr;
  const uint32_t *ksrc = from_labels ? seed_labels : keys;
  if (fast && seed_bits) {
    uint32_t iv[RX_P];
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      const uint32_t gyc = (uint32_t)min(max(gyb + r, 0), H - 1);
      // (a bit plane exists only for planes of fewer than 2^31 pixels: pixel indices fit 32 bits)
      const uint32_t ro = dims24 ? __umul24(gyc, (uint32_t)W) : gyc * (uint32_t)W;
      const uint32_t p = ro + (uint32_t)gxc0, ph_ = ro + (uint32_t)xh;
      const uint32_t nib = ksrc[p >> 5] >> (p & 31u);
      iv[r] = *reinterpret_cast<const uint32_t *>(img + ((unsigned long long)gyc * stride32 + (uint32_t)gxc0));
      halo[r] = (ksrc[ph_ >> 5] >> (ph_ & 31u)) & 1u;
      T[r][0] = nib & 1u; T[r][1] = nib & 2u; T[r][2] = nib & 4u; T[r][3] = nib & 8u;
    }
    patch_bases(iv, B, max_level);
    const uint32_t p = (dims24 ? __umul24((uint32_t)gy_halo, (uint32_t)W) : (uint32_t)gy_halo * (uint32_t)W) + (uint32_t)gxc0;
    const uint32_t nib = ksrc[p >> 5] >> (p & 31u);
    halo_row = u32x4_t{nib & 1u, nib & 2u, nib & 4u, nib & 8u};
  } else if (fast) {
    u32x4_t kv[RX_P];
    uint32_t iv[RX_P];
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      const int gyc = min(max(gyb + r, 0), H - 1);
      iv[r] = *reinterpret_cast<const uint32_t *>(img + ((unsigned long long)(uint32_t)gyc * stride32 + (uint32_t)gxc0));
      if (PERSIST && !(use_list & 1)) {      // (PERSIST: use_list carries the tuning build's A/B bits -- 1: plain stamp accesses, 2: slow polls)
        coh_load4(kv[r], ksrc + (size_t)gyc * W + gxc0);
        coh_load1(halo[r], ksrc + (size_t)gyc * W + xh);
      } else {
        kv[r] = *reinterpret_cast<const u32x4_t *>(ksrc + (size_t)gyc * W + gxc0);
        halo[r] = ksrc[(size_t)gyc * W + xh];
      }
    }
    if (PERSIST && !(use_list & 1)) {
      coh_load4(halo_row, ksrc + (size_t)gy_halo * W + gxc0);
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(kv[0]), "+v"(kv[1]), "+v"(kv[2]), "+v"(kv[3]), "+v"(halo_row), "+v"(halo[0]), "+v"(halo[1]), "+v"(halo[2]), "+v"(halo[3])
                   :
                   : "memory");
    } else {
      halo_row = *reinterpret_cast<const u32x4_t *>(ksrc + (size_t)gy_halo * W + gxc0);
    }
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      T[r][0] = kv[r].x; T[r][1] = kv[r].y; T[r][2] = kv[r].z; T[r][3] = kv[r].w;
    }
    patch_bases(iv, B, max_level);
  } else {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      const int gyc = min(max(gyb + r, 0), H - 1);
#pragma unroll
      for (int c = 0; c < RX_P; ++c) {
        const int gxc = min(max(gx0 + c, 0), W - 1);
        T[r][c] = plane_or_bit(ksrc, (size_t)gyc * W + gxc, seed_bits);
        const uint32_t v = img[pad ? padded_img_index(gyc, gxc, W, SH, img_stride) : (size_t)gyc * img_stride + gxc];
        B[r][c] = v <= max_level ? ((v << 24) | 1u) : KEY_INF;
      }
      halo[r] = plane_or_bit(ksrc, (size_t)gyc * W + xh, seed_bits);
    }
    halo_row.x = plane_or_bit(ksrc, (size_t)gy_halo * W + min(max(gx0 + 0, 0), W - 1), seed_bits);
    halo_row.y = plane_or_bit(ksrc, (size_t)gy_halo * W + min(max(gx0 + 1, 0), W - 1), seed_bits);
    halo_row.z = plane_or_bit(ksrc, (size_t)gy_halo * W + min(max(gx0 + 2, 0), W - 1), seed_bits);
    halo_row.w = plane_or_bit(ksrc, (size_t)gy_halo * W + min(max(gx0 + 3, 0), W - 1), seed_bits);
  }
  if (from_labels) {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
#pragma unroll
      for (int c = 0; c < RX_P; ++c) T[r][c] = T[r][c] ? 0u : KEY_INF;
      halo[r] = halo[r] ? 0u : KEY_INF;
    }
    halo_row.x = halo_row.x ? 0u : KEY_INF; halo_row.y = halo_row.y ? 0u : KEY_INF;
    halo_row.z = halo_row.z ? 0u : KEY_INF; halo_row.w = halo_row.w ? 0u : KEY_INF;
  }
  // (bases: only interior pixels with img <= max level can ever be flooded (lib.rs:220-224); everything else, and every
  // seed (stamp 0 < base), is pinned at its current stamp: b = t -- patch_bases above, the border masks and the min below)
  // Workgroup uniform: the tile and its halo ring lie strictly inside the image (and the image is not a stack of
  // slices) -- every pixel is in the plane and interior, none of the masks below can bite.  These kernels are VALU-bound
  // (VALUBusy 76-84 %, profiles/), and the masks were ~8 ops per pixel of a tile run's ~60.
  const bool inner = SEAM != 2 && SH == H && x0 >= 1 && x0 + TW <= W - 1 && y0 >= 1 && y0 + TH <= H - 1;
  if (!inner) {
    // row inside its slice: the four rows of a patch are all above the plane or all from row 0 on
    const int ry0 = (SH == H || gyb < 0) ? gyb : gyb % SH;
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      const int gy = gyb + r;
      const int ry = ry0 + r >= SH ? ry0 + r - SH : ry0 + r;
      const bool row_ok = gy >= 0 && gy < H, row_int = ry >= 1 && ry < SH - 1 && gy < H;
#pragma unroll
      for (int c = 0; c < RX_P; ++c) {
        const int gx = gx0 + c;
        if (!(row_ok && gx >= 0 && gx < W)) T[r][c] = KEY_INF;
        if (!(row_int && gx >= 1 && gx < W - 1)) B[r][c] = KEY_INF;
      }
      if (!(row_ok && xh_ok)) halo[r] = KEY_INF;
    }
    const bool ok = gy_halo_raw >= 0 && gy_halo_raw < H;
    if (!(ok && gx0 + 0 >= 0 && gx0 + 0 < W)) halo_row.x = KEY_INF;
    if (!(ok && gx0 + 1 >= 0 && gx0 + 1 < W)) halo_row.y = KEY_INF;
    if (!(ok && gx0 + 2 >= 0 && gx0 + 2 < W)) halo_row.z = KEY_INF;
    if (!(ok && gx0 + 3 >= 0 && gx0 + 3 < W)) halo_row.w = KEY_INF;
  }
#pragma unroll
  for (int r = 0; r < RX_P; ++r)
#pragma unroll
    for (int c = 0; c < RX_P; ++c) B[r][c] = min(B[r][c], T[r][c]);
  // The columns left / right of the patch, one register per patch row, PERSISTENT: a sweep refreshes them with one DPP
  // wave shift each, whose `old` operand is the register itself -- lane 0 (lane 63) has no neighbour lane and keeps what
  // it holds, the tile's halo column, for the whole tile run.  (Passing the halo as `old` every time cost a v_mov per
  // shift to set the destination up: 16 of the ~80 vector instructions of a sweep.)
  // (SPLIT: lane 32 is the first lane of the lower band and lane 31 the last of the upper one: the shift hands them a
  // value from the other band, which a select replaces with their halo column again)
  uint32_t Lh[RX_P], Rh[RX_P];
#pragma unroll
  for (int r = 0; r < RX_P; ++r) { Lh[r] = halo[r]; Rh[r] = halo[r]; }
  // the tile's halo column as every lane of the row sees it: the register of the row's first / last lane
  auto row_first = [&](uint32_t v) -> uint32_t {
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    if (!SPLIT) return a;
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    return lane < 32 ? a : b;
  };
  auto row_last = [&](uint32_t v) -> uint32_t {
    const uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
    if (!SPLIT) return b;
    const uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 31);
    return lane < 32 ? a : b;
  };
  auto refresh_columns = [&]() {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      Lh[r] = lane_left(Lh[r], T[r][3]);
      Rh[r] = lane_right(Rh[r], T[r][0]);
      if (SPLIT) { Lh[r] = lane == 32 ? halo[r] : Lh[r]; Rh[r] = lane == 31 ? halo[r] : Rh[r]; }
      if (SEAM == 2) { Lh[r] = (lane & 1) ? Lh[r] : halo[r]; Rh[r] = (lane & 1) ? halo[r] : Rh[r]; }
    }
  };
  {
    if (band == 0) *reinterpret_cast<u32x4_t *>(&sRow[0][xl * RX_P]) = halo_row;
    if (band == NB - 1) *reinterpret_cast<u32x4_t *>(&sRow[2 * NB + 1][xl * RX_P]) = halo_row;
    *reinterpret_cast<u32x4_t *>(&sRow[1 + 2 * band][xl * RX_P]) = u32x4_t{T[0][0], T[0][1], T[0][2], T[0][3]};
    *reinterpret_cast<u32x4_t *>(&sRow[2 + 2 * band][xl * RX_P]) = u32x4_t{T[3][0], T[3][1], T[3][2], T[3][3]};
  }
  if (band == 0) *reinterpret_cast<u32x4_t *>(&sInitRow[0][xl * RX_P]) = u32x4_t{T[0][0], T[0][1], T[0][2], T[0][3]};
  if (band == NB - 1) *reinterpret_cast<u32x4_t *>(&sInitRow[1][xl * RX_P]) = u32x4_t{T[3][0], T[3][1], T[3][2], T[3][3]};
  uint32_t init_col[RX_P];          // SEAM 2: this lane's outer column as loaded (every lane has one)
  if (SEAM == 2) {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) init_col[r] = T[r][(lane & 1) ? 3 : 0];
  } else if (xl == 0 || xl == LX - 1) {
#pragma unroll
    for (int r = 0; r < RX_P; ++r) sInitCol[xl == 0 ? 0 : 1][band * RX_P + r] = T[r][xl == 0 ? 0 : 3];
  }
  __syncthreads();
  WS_STAMP(1);
  WS_QPHASE(0);

  // ---- relaxation ---------------------------------------------------------------------------
  // One round = three free-running sweeps (down, right, up), a barrier that publishes the band
  // boundary rows, then ONE checked sweep (left) on fresh neighbours.  The fixpoint is reached
  // exactly when that checked sweep changes nothing in the whole tile: every pixel has then been
  // evaluated against final neighbour values.  Only the checked sweep pays for change tracking, and
  // a tile that was already converged leaves after 4 sweeps instead of 8.
  if (!from_labels) {                // (kernel uniform: the pass that creates the stamp plane writes every patch anyway)
    uint64_t sum_before = 0;         // stamps only ever decrease: a 64-bit patch sum tells "changed" exactly
#pragma unroll
    for (int r = 0; r < RX_P; ++r)
#pragma unroll
      for (int c = 0; c < RX_P; ++c) sum_before += T[r][c];
    s_sum[tid] = sum_before;
  }
  uint32_t iters = 0;
  WS_ACC_DECL;
  bool unfinished = max_iters == 0;      // left before the checked sweep came back clean (round cap)
  // the three free sweeps of a round, then the band boundary rows are published
  auto free_sweeps = [&](uint32_t round) {
    bool untracked = false;
    uint32_t up[RX_P], dn[RX_P];
    {
      const u32x4_t up4 = *reinterpret_cast<const u32x4_t *>(&sRow[2 * band][xl * RX_P]);
      const u32x4_t dn4 = *reinterpret_cast<const u32x4_t *>(&sRow[2 * band + 3][xl * RX_P]);
      up[0] = up4.x; up[1] = up4.y; up[2] = up4.z; up[3] = up4.w;
      dn[0] = dn4.x; dn[1] = dn4.y; dn[2] = dn4.z; dn[3] = dn4.w;
    }
    if (!SCAN) {
      refresh_columns();
      sweep_rows<false, true>(T, B, up, dn, Lh, Rh, untracked);       // down
      refresh_columns();
      sweep_cols<false, true>(T, B, up, dn, Lh, Rh, untracked);       // right
      refresh_columns();
      sweep_rows<false, false>(T, B, up, dn, Lh, Rh, untracked);      // up
    } else {
      // The long-range variant: exact scans right, left, then down and up in one phase, instead of the three sweeps (which move a stamp by one
      // patch; with them as well a round cost a quarter more and the passes were no fewer: gpurun_out/r2w/skipfree.log).
      // The checked sweep that follows still sees every pixel's four neighbours: the exit test is the same.
      WS_ACC_T0;
#pragma unroll
      for (int r = 0; r < RX_P; ++r) scan_row<false, true, LX>(T[r], B[r], row_first(Lh[r]), xl, untracked);
#pragma unroll
      for (int r = 0; r < RX_P; ++r) scan_row<false, false, LX>(T[r], B[r], row_last(Rh[r]), xl, untracked);
      WS_ACC(0);
      scan_cols_both<NB, TW, SPLIT>(T, B, &sFn[0][0][0][0], sRow[0], sRow[2 * NB + 1], band, xl);
      WS_ACC(1);
    }
    *reinterpret_cast<u32x4_t *>(&sRow[1 + 2 * band][xl * RX_P]) = u32x4_t{T[0][0], T[0][1], T[0][2], T[0][3]};
    *reinterpret_cast<u32x4_t *>(&sRow[2 + 2 * band][xl * RX_P]) = u32x4_t{T[3][0], T[3][1], T[3][2], T[3][3]};
    __syncthreads();
  };
  // LITE (passes >= 2: few pixels are still wrong, many flagged tiles need no change at all): the checked sweep comes
  // FIRST and the free sweeps after it -- a tile that is already a fixpoint leaves after one sweep instead of four.
  // Same sequence of sweeps as the other order minus the first three; the exit test is the same.
  for (; max_iters != 0;) {
    ++iters;
    // (chunk == 3, pass 0 of a seam-repair transform: after two full rounds a tile of the bench field has two or three stamps
    // left to settle -- from the third round on a round is its checked sweep alone, one sweep to settle them and one to see
    // that nothing moves, instead of a round of four that finds nothing to do)
    if (!LITE && !(chunk == 3 && iters > 2)) free_sweeps(iters);
    bool changed = false;
    uint32_t up[RX_P], dn[RX_P];
    {
      const u32x4_t up4 = *reinterpret_cast<const u32x4_t *>(&sRow[2 * band][xl * RX_P]);
      const u32x4_t dn4 = *reinterpret_cast<const u32x4_t *>(&sRow[2 * band + 3][xl * RX_P]);
      up[0] = up4.x; up[1] = up4.y; up[2] = up4.z; up[3] = up4.w;
      dn[0] = dn4.x; dn[1] = dn4.y; dn[2] = dn4.z; dn[3] = dn4.w;
    }
    WS_ACC_T0;
    refresh_columns();
    sweep_cols<true, false>(T, B, up, dn, Lh, Rh, changed);         // left, checked
    WS_ACC(2);
    const uint32_t slot = (iters - 1) % 3;
    if (__builtin_amdgcn_ballot_w64(changed) != 0) {
      // a neighbour band reads these rows only if another round follows, i.e. only if someone changed
      *reinterpret_cast<u32x4_t *>(&sRow[1 + 2 * band][xl * RX_P]) = u32x4_t{T[0][0], T[0][1], T[0][2], T[0][3]};
      *reinterpret_cast<u32x4_t *>(&sRow[2 + 2 * band][xl * RX_P]) = u32x4_t{T[3][0], T[3][1], T[3][2], T[3][3]};
      if (lane == 0) s_flag[slot] = 1;
    }
    __syncthreads();
    const bool again = s_flag[slot] != 0;
    if (tid == 0) s_flag[(slot + 2) % 3] = 0;
    if (!again) break;
    if (iters >= max_iters) { unfinished = true; break; }
    if (LITE) free_sweeps(iters);
  }
  bool any_lower = true;             // a pass that creates the stamp plane writes every patch
  if (!from_labels) {
    uint64_t sum_after = 0;
#pragma unroll
    for (int r = 0; r < RX_P; ++r)
#pragma unroll
      for (int c = 0; c < RX_P; ++c) sum_after += T[r][c];
    any_lower = sum_after != s_sum[tid];
  }
  WS_STAMP(2);
  WS_QPHASE(1);
  WS_ACC_STORE;
#ifdef WS_DIAG_STAMPS
  if (threadIdx.x == 0 && g_diag) g_diag[(size_t)blockIdx.x * 8 + 4] = iters;
#endif

  // ---- write back the patches that changed (16 B per lane and row), ring-carry check, edge flags
  uint32_t e = 0, ovf = 0;
  if (any_lower) {
    const bool full_x = gx0 >= 0 && gx0 + RX_P <= W;
#pragma unroll
    for (int r = 0; r < RX_P; ++r) {
      const int gy = gyb + r;
      if (gy >= 0 && gy < H) {
        if (full_x) {
          if (PERSIST && !(use_list & 1)) coh_store4(keys + (size_t)gy * W + gx0, u32x4_t{T[r][0], T[r][1], T[r][2], T[r][3]});
          else *reinterpret_cast<u32x4_t *>(keys + (size_t)gy * W + gx0) = u32x4_t{T[r][0], T[r][1], T[r][2], T[r][3]};
        } else {
#pragma unroll
          for (int c = 0; c < RX_P; ++c) if (gx0 + c >= 0 && gx0 + c < W) keys[(size_t)gy * W + gx0 + c] = T[r][c];
        }
      }
      if (check_carry) {               // kernel uniform
#pragma unroll
        for (int c = 0; c < RX_P; ++c)   // a finite non-seed stamp with ring 0 can only come from a carry out of the ring field
          ovf |= (T[r][c] != 0u && T[r][c] < KEY_INF && (T[r][c] & RING_MASK) == 0u);
      }
    }
    e |= 16u;
    // A tile that stopped at the round cap is not a fixpoint of its own pixels: the four tiles of the
    // other grid that cover it re-examine all of it in the next pass.
    // (before a same-grid pass the tile marks ITSELF instead: word 0 of its stamps, below)
    if (unfinished && !write_same) e |= 15u;
    // bit q: a BORDER pixel of the tile inside quadrant q = 2*(lower half) + (right half) changed
    const uint32_t qbit = 1u << ((band >= NB / 2 ? 2 : 0) + (xl >= LX / 2 ? 1 : 0));
    // (bits 6 .. 9: the same changes by SIDE -- top row, bottom row, left column, right column -- for the passes that
    // stay on one grid, where a side matters to exactly one neighbour)
    // (SEAM 1: what a band changes in its first and last four columns, where a vertical seam runs, is the business of the
    // strip that follows -- it looks at those columns again, in every row -- and raises no flag here: with them a band's
    // first / last row "changed" in a quarter of the bands, whatever their height, for the two seam pixels at its corners;
    // tools/sim_tile_schedule.c, SIM_REPAIR: 55 % of the tiles flagged with them, 41 % without, 9 % with bands of 16 rows)
    const bool row_counts = SEAM != 1 || !((lane == 0 && x0 > 0) || (lane == 63 && x0 + TW < W));
    // A changed border pixel only MATTERS to the tile across the border when it can lower the pixel it touches there:
    // new stamp + 1 < that pixel's stamp -- which this tile holds, as its halo.  (The halo is as old as the tile's load: the
    // pixel can only have fallen since, so the test errs on the side of flagging.)  On smooth maps a third to a half of the
    // late tile runs changed nothing at all (tools/sim_tile_schedule.c, SIM_CHANGED): flagged by a neighbour whose front
    // had not caught up with theirs.
    auto matters = [](uint32_t before, uint32_t now, uint32_t across) { return before != now && now + 1u < across; };
    auto least = [](uint32_t before, uint32_t now, uint32_t across) { return before != now && now + 1u < across ? now : 0xFFFFFFFFu; };      // (PERSIST == 2: the queue's order)
    if (band == 0) {
      const u32x4_t o = *reinterpret_cast<const u32x4_t *>(&sInitRow[0][xl * RX_P]);
      const u32x4_t a = *reinterpret_cast<const u32x4_t *>(&sRow[0][xl * RX_P]);      // the halo row above, as loaded
      if (matters(o.x, T[0][0], a.x) || matters(o.y, T[0][1], a.y) || matters(o.z, T[0][2], a.z) || matters(o.w, T[0][3], a.w)) e |= qbit | (row_counts ? 64u : 0u);
      if (PERSIST == 2 && (e & 64u)) atomicMin(&s_sidemin[0], min(min(least(o.x, T[0][0], a.x), least(o.y, T[0][1], a.y)), min(least(o.z, T[0][2], a.z), least(o.w, T[0][3], a.w))));
    }
    if (band == NB - 1) {
      const u32x4_t o = *reinterpret_cast<const u32x4_t *>(&sInitRow[1][xl * RX_P]);
      const u32x4_t a = *reinterpret_cast<const u32x4_t *>(&sRow[2 * NB + 1][xl * RX_P]);      // the halo row below
      if (matters(o.x, T[3][0], a.x) || matters(o.y, T[3][1], a.y) || matters(o.z, T[3][2], a.z) || matters(o.w, T[3][3], a.w)) e |= qbit | (row_counts ? 128u : 0u);
      if (PERSIST == 2 && (e & 128u)) atomicMin(&s_sidemin[1], min(min(least(o.x, T[3][0], a.x), least(o.y, T[3][1], a.y)), min(least(o.z, T[3][2], a.z), least(o.w, T[3][3], a.w))));
    }
    if (SEAM == 2) {
      // a lane pair raises its own flags: the anchored tile that holds this lane's columns (left of the seam for the even
      // lane, right of it for the odd one), and that tile's neighbour above / below when the slice's first / last row changed
      bool col = false;
#pragma unroll
      for (int r = 0; r < RX_P; ++r) col |= matters(init_col[r], T[r][(lane & 1) ? 3 : 0], (lane & 1) ? Rh[r] : Lh[r]);      // (Lh / Rh of these lanes: the halo column)
      col |= unfinished;      // stopped at the round cap: the tiles on both sides of every seam of this slice look again
      const int fx = seam_x / SEAM_PX - 1 + (lane & 1);
      const uint32_t mark = pass + 1;
      if (seam_x < W) {      // (anchored tile rows are SEAM_PY pixel rows; a slice lies in one of them)
        if (col && gyb >= 0 && gyb < H) { stamps_cur[((size_t)(gyb / SEAM_PY) * otherX + fx) * 4 + 3] = mark; e |= 1u; }
        if ((e & 64u) && y0 > 0) { stamps_cur[((size_t)((y0 - 1) / SEAM_PY) * otherX + fx) * 4 + 3] = mark; e |= 1u; }
        if ((e & 128u) && y0 + TH < H) { stamps_cur[((size_t)((y0 + TH) / SEAM_PY) * otherX + fx) * 4 + 3] = mark; e |= 1u; }
      }
      e &= 1u | 16u;
    } else if (xl == 0 || xl == LX - 1) {
#pragma unroll
      for (int r = 0; r < RX_P; ++r)
        if (matters(sInitCol[xl == 0 ? 0 : 1][band * RX_P + r], T[r][xl == 0 ? 0 : 3], xl == 0 ? Lh[r] : Rh[r])) e |= qbit | (xl == 0 ? 256u : 512u);      // (Lh of a row's first lane, Rh of its last: the halo column)
      if (PERSIST == 2 && (e & (256u | 512u))) {
        uint32_t m = 0xFFFFFFFFu;
#pragma unroll
        for (int r = 0; r < RX_P; ++r) m = min(m, least(sInitCol[xl == 0 ? 0 : 1][band * RX_P + r], T[r][xl == 0 ? 0 : 3], xl == 0 ? Lh[r] : Rh[r]));
        atomicMin(&s_sidemin[xl == 0 ? 2 : 3], m);
      }
    }
  }
  // (SEAM 2, a slice that stopped at its round cap: also the lanes that changed nothing ask for their tile's re-run)
  if (SEAM == 2 && unfinished && !any_lower && seam_x < W && gyb >= 0 && gyb < H) {
    stamps_cur[((size_t)(gyb / SEAM_PY) * otherX + (seam_x / SEAM_PX - 1 + (lane & 1))) * 4 + 3] = pass + 1;
    e |= 1u;
  }
  // PERSIST: every wave's stamps have left for memory before the workgroup's barrier, and so before wave 0 flags a neighbour
  if (PERSIST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ovf) atomicExch(pf.overflow, 1u);      // never taken on sane inputs
  if (unfinished && (write_same || SEAM == 1 || chunk == 3)) e |= 32u;      // (chunk == 3: pass 0 of a seam-repair transform)
  if (e) atomicOr(&s_edges, e);
  __syncthreads();
  if (tid == 0) {
    const uint32_t ed = s_edges;
    const uint32_t stripe = (blockIdx.x % NSTRIPE) * STRIPE_STRIDE;
    if (SEAM) {
      // SEAM 1: the tile above the seam holds the pixels over this tile's first row, the tile below those under its last;
      // what its first / last COLUMN changed lies inside a vertical strip, which runs after the bands.  (SEAM 2: the lanes
      // have stored their flags themselves.)
      bool any = SEAM == 2 && (ed & 1u) != 0u;
      if (SEAM == 1) {      // (bit 5: the band stopped at its round cap)
        if (ed & (64u | 32u)) { stamps_cur[((size_t)tile_y * otherX + tile_x) * 4 + 3] = pass + 1; any = true; }
        if (ed & (128u | 32u)) { stamps_cur[((size_t)(tile_y + 1) * otherX + tile_x) * 4 + 3] = pass + 1; any = true; }
      }
      if (any) pf.edge_changed[(pass % COUNTER_RING) * FLAG_SLOT + stripe] = 1u;
      if (ed) pf.any_change[stripe] = 1u;
    } else if (ed) {
      const size_t t = (size_t)tile_y * tilesX + tile_x;
      // pass 0 of a seam-repair transform (chunk == 3) that stopped at its round cap: nobody reads this pass's own stamps
      // (the bands and strips look at every seam anyway), so the tile asks for its re-run where pass 2 will look -- the
      // word of the OTHER stamp array that the bands and strips use for the same purpose
      // (pass 2 runs anchored tile (x, y) for word 3 of entry (x, y) or word 2 of entry (x + 1, y): the first is the bands'
      // and strips', the second this one's, so that they can tell a tile pass 0 gave up on from one a band has flagged)
      if (!CHUNKED && chunk == 3 && (ed & 32u)) const_cast<uint32_t *>(stamps_prev)[((size_t)tile_y * otherX + tile_x + 1) * 4 + 2] = pass + 2;
      if (write_same) {      // the next pass runs on this grid (relax_todo, read_same)
        if (ed & 33u) stamps_cur[t * 4 + 0] = (pass + 1) | (ed & 1u ? ST_BORDER : 0u) | (ed & 32u ? ST_SELF : 0u);
        if (ed & 2u) stamps_cur[t * 4 + 1] = (pass + 1) | ST_BORDER;
        if (ed & 4u) stamps_cur[t * 4 + 2] = (pass + 1) | ST_BORDER;
        if (ed & 8u) stamps_cur[t * 4 + 3] = (pass + 1) | ST_BORDER;
      } else {
        if (ed & 1u) stamps_cur[t * 4 + 0] = pass + 1;
        if (ed & 2u) stamps_cur[t * 4 + 1] = pass + 1;
        if (ed & 4u) stamps_cur[t * 4 + 2] = pass + 1;
        if (ed & 8u) stamps_cur[t * 4 + 3] = pass + 1;
      }
      // plain, idempotent stores into striped words: no same-address atomics on the tile path
      if (ed & 47u) pf.edge_changed[(pass % COUNTER_RING) * FLAG_SLOT + stripe] = 1u;
      if (append_next) {
        // The next pass's tile list is written by the tiles that cause its entries (no k_relax_list launch between two
        // passes: 5 us of every ~45).  A tighter rule than relax_todo's same-grid test (which only has the quadrant
        // stamps): a changed top-row pixel matters to the tile above and to nobody else, and so on round the tile -- a tenth
        // fewer tile runs on smooth maps than "every neighbour that touches the quadrant"; I go on myself if I stopped at
        // the round cap.  The candidates only go to LDS here: the workgroup hands them in together (append_flush).
        const bool want[5] = {(ed & 32u) != 0u, (ed & 64u) != 0u && tile_y > 0, (ed & 128u) != 0u && tile_y + 1 < tilesY,
                              (ed & 256u) != 0u && tile_x > 0, (ed & 512u) != 0u && tile_x + 1 < tilesX};
        const uint32_t who[5] = {(uint32_t)t, (uint32_t)(t - tilesX), (uint32_t)(t + tilesX), (uint32_t)(t - 1), (uint32_t)(t + 1)};
        uint32_t n = s_ncand;
        if (PERSIST == 2) {
          // the bucket of what I announce: the level of the smallest stamp that matters across that side; for myself (I
          // stopped at the round cap) the lowest of them and of the bucket I ran from
          uint32_t bk[5];
          bk[0] = s_qbucket;
#pragma unroll
          for (int k = 1; k < 5; ++k) {
            bk[k] = min(s_sidemin[k - 1] >> pq_shift, (uint32_t)(PQ_B - 1));
            if (want[k]) bk[0] = min(bk[0], bk[k]);
          }
#pragma unroll
          for (int k = 0; k < 5; ++k)
            if (want[k]) s_cand[n++] = who[k] | (bk[k] << 24);
        } else {
#pragma unroll
          for (int k = 0; k < 5; ++k)
            if (want[k]) s_cand[n++] = who[k];
        }
        s_ncand = n;
      }
      pf.any_change[stripe] = 1u;
    }
    if (pf.stats) {            // profiling only: striped counters, one per 64-byte line
      atomicAdd(&pf.stats[stripe], (uint32_t)(TW * TH / 2048));      // in quarter tiles: a 256 x 8 band is one, every other tile four (ws_segment.hip divides)
      atomicAdd(&pf.stats[FLAG_SLOT + stripe], iters);
    }
  }
  WS_STAMP(3);
  WS_QPHASE(2);
  if (PERSIST == 2) {
    __syncthreads();      // thread 0's candidates (s_cand, s_ncand: the append_next block above) are there
    if (tid < 64) {
      const uint32_t self = (uint32_t)tile, n = s_ncand;
      // my own entry (a run that stopped at its round cap) is lane 0's first business: its bit, then "not running any more"
      // in one more exchange that tells me what was announced while I ran
      const bool mine = (uint32_t)lane < n;
      const uint32_t cw = mine ? s_cand[lane] : 0u;
      const uint32_t cand = cw & 0x00FFFFFFu, cb = cw >> 24;      // (relax_pass: the queue is not used on planes of 2^24 tiles)
      const bool other = mine && cand != self;
      // Hand-off: the announced tile of the lowest bucket, if that is no higher than the bucket I ran from (the front I am
      // following), is MY next tile -- one exchange "not running -> running" instead of its bit, its count, some worker's
      // look, claim and exchange: five memory round trips off every hop of a flood that is a chain of tile runs.
      // (Asking the counts instead -- "nothing waits below it", a look issued before the stores -- cost more than it found:
      // one more request per run to the line every worker's counts live on, 4.98 -> 5.32 ms at correlation 64 px.)
      uint32_t pick = other && (cb <= s_qbucket || (use_list & 16)) && !(use_list & 8) ? (cb << 8) | (uint32_t)lane : 0xFFFFu;
#pragma unroll
      for (int k = 1; k < 8; k <<= 1) pick = min(pick, (uint32_t)__shfl_xor((int)pick, k, 64));      // (candidates sit in lanes 0 .. 4)
      pick = (uint32_t)__shfl((int)pick, 0, 64);
      const bool hand = pick != 0xFFFFu && (pick & 0xFFu) == (uint32_t)lane;
      uint32_t old = 0;
      bool taken = false;
      if (hand) {
        old = atomicMax(&q_state[cand], PQ_RUNNING);
        taken = !(old & PQ_RUNNING);      // idle (old == 0: mine to count) or queued (its bit goes stale): it is mine now
      }
      if (other && !taken) old = atomicOr(&q_state[cand], 1u << cb);
      // idle: mine to queue (and to count); queued in a higher bucket: mine to queue lower, and its old bit goes;
      // queued at or below mine: nothing; running: announced, its run's end queues it
      bool push = other && !taken && !(old & PQ_RUNNING) && (old & ((2u << cb) - 1u)) == 0u;
      const bool fresh = (push || taken) && old == 0u;
      if (taken) { s_handoff = 1u; s_qtile = cand + 1u; s_qbucket = cb; }
      uint32_t pb = cb, pt = cand;
      uint32_t left = 0;
      if (lane == 63) {      // (never a candidate's lane: a run announces five tiles at most)
        const uint32_t c0 = s_cand[0];
        if (n != 0u && (c0 & 0x00FFFFFFu) == self) atomicOr(&q_state[self], 1u << (c0 >> 24));
        left = atomicAnd(&q_state[self], ~PQ_RUNNING) & ~PQ_RUNNING;
        if (left) { push = true; pb = (uint32_t)__builtin_ctz(left); pt = self; }
      }
      const uint32_t n_fresh = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(fresh));
      if (lane == 63) {
        // counted before anyone can take them; my own run leaves the count in the same add
        const int net = (int)n_fresh - (left ? 0 : 1);
        if (net > 0) atomicAdd(q_pending, (uint32_t)net);
        else if (net < 0 && atomicSub(q_pending, 1u) == 1u) {
          __hip_atomic_store(q_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(pq_avail + PQ_B, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        s_ncand = 0;
#ifdef WS_TUNING
        atomicAdd(tile_list + RLQ_RUNS, 1u);
#endif
      }
      if (push) {
        const uint32_t bit = 1u << (pt & 31u);
        if (!(atomicOr(pq_bits + (size_t)pb * pq_bw + (pt >> 5), bit) & bit)) atomicAdd(pq_avail + pb, 1u);
        if (pt != self && old != 0u) {      // lowered: the bit of its former bucket
          const uint32_t ob = (uint32_t)__builtin_ctz(old);
          if (atomicAnd(pq_bits + (size_t)ob * pq_bw + (pt >> 5), ~bit) & bit) atomicSub(pq_avail + ob, 1u);
        }
      }
    }
    WS_QPHASE(3);
    continue;
  }
  if (PERSIST) {
    __syncthreads();      // thread 0's candidates (s_cand, s_ncand: the append_next block above) are there
    if (tid < 64) {
      const uint32_t self = (uint32_t)tile, n = s_ncand;
      const bool mine = (uint32_t)lane < n;
      const uint32_t cand = mine ? s_cand[lane] : 0u;
      // idle -> queued: mine to push; queued already: nothing; running: flagged, it queues itself when it ends
      const bool fresh = mine && atomicOr(&q_state[cand], 1u) == 0u;
      const unsigned long long fm = __builtin_amdgcn_ballot_w64(fresh);
      const uint32_t nf = (uint32_t)__popcll(fm);
      // my run is over: running -> idle, or -> queued if somebody (I myself, at my round cap) flagged me meanwhile
      uint32_t again = 0;
      if (lane == 0) again = atomicAnd(&q_state[self], ~2u) & 1u;
      again = (uint32_t)__shfl((int)again, 0, 64);
      const uint32_t total = nf + again;
      if (total) {
        uint32_t at = 0;
        if (lane == 0) {      // counted before anyone can take them; my own run leaves the count in the same add
          if (total != 1u) atomicAdd(q_pending, total - 1u);
          at = atomicAdd(q_tail, total);
        }
        at = (uint32_t)__shfl((int)at, 0, 64);
        if (fresh) {
          const uint32_t idx = at + (uint32_t)__popcll(fm & ((1ull << lane) - 1ull));
          __hip_atomic_store(q_ring + (idx % list_cap), ((unsigned long long)(idx + 1u) << 32) | cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (lane == 0 && again) {
          const uint32_t idx = at + nf;
          __hip_atomic_store(q_ring + (idx % list_cap), ((unsigned long long)(idx + 1u) << 32) | self, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (lane == 0) {
        s_ncand = 0;
        // nothing queued by me and my run is over: was mine the last tile queued or running?
        if (total == 0u && atomicSub(q_pending, 1u) == 1u) __hip_atomic_store(q_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef WS_TUNING
        atomicAdd(tile_list + RLQ_RUNS, 1u);
#endif
      }
    }
    WS_QPHASE(3);
    continue;
  }
  // next tile of the chunk: every wave is past its last read of the shared arrays (barrier above)
  if (!CHUNKED) break;
  if (use_list) {
    entry = s_next[runs_done & 1u];      // (written before this run's first barrier)
    ++runs_done;
    const bool last = entry >= n_entries;
    // wave 0 hands the candidates in: when this was the workgroup's last tile, or when another tile's five might not fit
    if (append_next && tid < 64 && (last || s_ncand > RX_CAND - 5u)) append_flush();
    if (last) break;
  } else {
    todo &= todo - 1;
    if (todo == 0) break;
  }
  }
}

constexpr int RX_NW = 8;   // 512 threads: tile 256 x 32

// The tiles that have to run in `pass`, compacted: tile_list[pass & 3] = how many, entries from tile_list[RL_HDR + (pass & 1) *
// list_cap] on (any order).  It also clears the tiles' "queued for pass" words, which the passes that append their
// successors' lists themselves (k_relax, append_next) exchange: every transform that reaches those passes comes through here.
// Same test as relax_todo; one atomicAdd per wave that found any.  Worth its own launch only when few tiles run: on a
// smooth map a pass moves the flood fronts by one tile, a few hundred tiles out of thousands, and the chunked launch
// (a workgroup per four tiles, most of them idle, some with two busy ones to run back to back) took twice as long as
// the tiles themselves.
// first pass that runs on the grid of the pass before it (odd; see relax_pass).  8192^2 smooth maps, correlation length
// 64 / 256 px: 436 -> 260 and 516 -> 292 passes, 16.8 -> 12.6 and 11.0 -> 7.0 ms (gpurun_out/r2k, bit-exact)
constexpr uint32_t RX_SAME_GRID_FROM = 7;
constexpr uint32_t RX_LIST_FROM_PASS = 6;      // the bench field has converged by then (its passes 4 and 5 find nothing to do)
constexpr unsigned RX_LIST_GRID = 512;       // two workgroups of the scan variant per CU: all resident, the tickets share the list out

template <int TW, int TH>
__global__ __launch_bounds__(256) void k_relax_list(int H, int W, int tilesX, int tilesY, int otherX, int otherY, int shifted,
                                                    uint32_t pass, const uint32_t *__restrict__ stamps_prev, uint32_t *tile_list,
                                                    int read_same, uint32_t list_cap) {
  const int lane = threadIdx.x & 63;
  const int first = (int)((blockIdx.x * blockDim.x + threadIdx.x) & ~63u);      // this wave's 64 consecutive tiles
  if ((uint32_t)(first + lane) <= list_cap) tile_list[RL_HDR + 2 * (size_t)list_cap + first + lane] = 0u;      // queued marks (and the dummy slot)
  const unsigned long long todo = relax_todo<TW, TH>(first, 1, 64, H, W, tilesX, tilesY, otherX, otherY, shifted, pass, stamps_prev, read_same);
  if (todo == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&tile_list[pass & 3u], (uint32_t)__popcll(todo));
  base = __shfl(base, 0, 64);
  if ((todo >> lane) & 1ull)
    tile_list[RL_HDR + (pass & 1u) * (size_t)list_cap + base + __popcll(todo & ((1ull << lane) - 1ull))] = (uint32_t)(first + lane);
}

// The first pass on the 128 x 64 grid (RX_SAME_GRID_FROM): its tile list from the edge stamps that the pass before it left
// on the 256 x 32 grid.  An equation can only be left violated inside a tile that stopped at its round cap, or next to a
// border pixel that a tile changed: every new tile that touches an old tile with ANY stamp word of that pass (the old
// tile's rectangle grown by one pixel) runs -- a superset of the tiles relax_todo would pick, and running a tile that has
// nothing to do changes nothing.
template <int TW, int TH, int OTW, int OTH>
__global__ __launch_bounds__(256) void k_relax_list_regrid(int H, int W, int tilesX, int tilesY, int oldX, int oldY, uint32_t pass,
                                                           const uint32_t *__restrict__ stamps_prev, uint32_t *tile_list, uint32_t list_cap,
                                                           int persist) {
  // persist: the list is the first filling of the persistent pass's queue (k_relax, PERSIST) -- (sequence number, tile) pairs
  // in the ring, the tiles' state words "queued", the tiles counted in [RLQ_PENDING]; the host has zeroed ring and counters
  const int lane = threadIdx.x & 63;
  const int first = (int)((blockIdx.x * blockDim.x + threadIdx.x) & ~63u);
  const int t = first + lane;
  bool run = false;
  if (t < tilesX * tilesY) {
    const int tx = t % tilesX, ty = t / tilesX;
    const int x_lo = max(tx * TW - 1, 0) / OTW, x_hi = min((tx * TW + TW) / OTW, oldX - 1);
    const int y_lo = max(ty * TH - 1, 0) / OTH, y_hi = min((ty * TH + TH) / OTH, oldY - 1);
    for (int oy = y_lo; oy <= y_hi; ++oy)
      for (int ox = x_lo; ox <= x_hi; ++ox) {
        const uint4 st = *reinterpret_cast<const uint4 *>(stamps_prev + ((size_t)oy * oldX + ox) * 4);
        run |= (st.x & ST_PASS) == pass || (st.y & ST_PASS) == pass || (st.z & ST_PASS) == pass || (st.w & ST_PASS) == pass;
      }
    run = run && tx * TW < W && ty * TH < H;
  }
  if ((uint32_t)t <= list_cap) tile_list[RL_HDR + 2 * (size_t)list_cap + t] = persist && run ? 1u : 0u;      // queued marks (and the dummy slot)
  const unsigned long long todo = __builtin_amdgcn_ballot_w64(run);
  if (todo == 0) return;
  uint32_t base = 0;
  if (lane == 0) {
    base = atomicAdd(&tile_list[persist ? RLQ_TAIL : (pass & 3u)], (uint32_t)__popcll(todo));      // (persist == 2: a count for the diagnostics)
    if (persist) atomicAdd(&tile_list[RLQ_PENDING], (uint32_t)__popcll(todo));
  }
  base = __shfl(base, 0, 64);
  const uint32_t idx = base + (uint32_t)__popcll(todo & ((1ull << lane) - 1ull));
  if (persist == 2) {      // the queue in flood order: everything starts in bucket 0 (these 64 tiles are two words of its bitmap, this wave's alone)
    uint32_t *avail = tile_list + pq_base(list_cap);
    if (lane == 0) atomicAdd(avail, (uint32_t)__popcll(todo));
    if ((lane & 31) == 0) (avail + PQ_HDR)[(first >> 5) + (lane >> 5)] = (uint32_t)(todo >> (lane & 32));
  } else if (run && persist) reinterpret_cast<unsigned long long *>(tile_list + RL_HDR)[idx] = ((unsigned long long)(idx + 1u) << 32) | (uint32_t)t;
  else if (run) tile_list[RL_HDR + (pass & 1u) * (size_t)list_cap + idx] = (uint32_t)t;
}

// Every tile of the plane, as the list of `pass`: the pass after the persistent one looks at each of them once (a tile that
// is at its fixpoint leaves after one checked sweep), whatever the queue did.
template <int TW, int TH>
__global__ __launch_bounds__(256) void k_relax_list_all(int H, int W, int tilesX, int tilesY, uint32_t pass, uint32_t *tile_list, uint32_t list_cap) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t == 0) { tile_list[pass & 3u] = (uint32_t)(tilesX * tilesY); tile_list[4 + (pass & 3u)] = 0u; }
  if ((uint32_t)t <= list_cap) tile_list[RL_HDR + 2 * (size_t)list_cap + t] = 0u;      // queued marks: the append protocol of the later passes starts clean
  if (t < tilesX * tilesY) tile_list[RL_HDR + (pass & 1u) * (size_t)list_cap + t] = (uint32_t)t;      // (every tile of this grid starts inside the plane)
}

// words of scratch relax_pass wants for its tile lists
// header, two entry arrays, queued marks (+ dummies); the persistent pass's buckets: counts (a line each), bitmaps
size_t relax_list_words(int h, int w) {
  const uint32_t tiles = (uint32_t)relax_tiles(h, w);
  return pq_base(tiles) + PQ_HDR + (size_t)PQ_B * pq_words_per_bucket(tiles);
}

// capacity of ONE of the two edge-stamp arrays: the shifted grid has one more row and column; the 128 x 64 grid of the
// same-grid passes has its own count
#ifndef WS_SPLIT_NW
#define WS_SPLIT_NW RX_NW
#endif
constexpr int RX_SNW = WS_SPLIT_NW;
constexpr int RX_STW = RX_TW / 2, RX_STH = 2 * RX_SNW * RX_P;      // tile of the SPLIT kernel: 128 x 64
// The queue in flood order runs on tiles of twice the height, 128 x 128 (sixteen waves): the launch is bound by the chain of
// tile runs along the floods, and a flood crosses half as many of these vertically.  The ordinary same-grid passes are
// slower on them (a pass lasts as long as its slowest tile); 8192^2 smooth maps, correlation 4 / 16 / 64 / 256 px, passes:
// 3.06 / 6.04 / 6.94 / 3.73 ms on 128 x 64, 3.58 / 7.23 / 7.39 / 3.48 on 128 x 128; queue: 3.78 / 7.2 / 5.12 / 4.09 against
// 3.69 / 6.75 / 4.69 / 3.65 (gpurun_out/r3am).
#ifndef WS_QUEUE_NW
#define WS_QUEUE_NW 16
#endif
constexpr int RX_QNW = WS_QUEUE_NW;
constexpr int RX_QTH = 2 * RX_QNW * RX_P;
size_t relax_tiles(int h, int w) {
  const int th = RX_NW * RX_P;
  const size_t a = (size_t)((w + RX_TW - 1) / RX_TW + 1) * ((h + th - 1) / th + 1);
  const size_t b = (size_t)((w + RX_STW - 1) / RX_STW + 1) * ((h + RX_STH - 1) / RX_STH + 1);
  return std::max(a, b);
}

// Row block of a tiled field: the caller has rewritten the plane's halo rows (row 0 and / or row h - 1).  Only tiles that
// hold those rows have anything new to look at: raise, in the stamp array that the EVEN pass `pass` reads (the shifted
// grid's), every quadrant flag of its first / last tile row, so that exactly the first / last tile row of the anchored
// grid runs in that pass; what they change spreads by the usual flags.  The stamp arrays must be zeroed first.
__global__ void k_flag_tile_rows(uint32_t *prev, int sx, int bottom_row, uint32_t pass, int halo_flags) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= sx * 4) return;
  if (halo_flags & 1) prev[i] = pass;
  if (halo_flags & 2) prev[(size_t)bottom_row * sx * 4 + i] = pass;
}

hipError_t block_flag_border_tiles(hipStream_t s, uint32_t *stamps, int h, int w, uint32_t pass, int halo_flags) {
  if ((pass & 1u) != 0 || h == 0 || w == 0) return hipErrorInvalidValue;
  const int th = RX_NW * RX_P;
  const int sx = (w + RX_TW - 1) / RX_TW + 1;
  const size_t cap = relax_tiles(h, w) * 4;
  // An anchored tile row t runs when row t or t + 1 of the shifted grid is flagged.  What has to run is every tile that
  // holds a pixel NEXT to a halo row, i.e. plane rows 1 (tile row 0: shifted row 0) and h - 2 -- which need not share a
  // tile with the halo row h - 1 itself (a block of 32 k + 1 rows: the halo row has a tile row of its own, and running
  // only that one, whose pixels are all pinned, repairs nothing).  Shifted row (h - 2) / th + 1 starts tile row
  // (h - 2) / th and, when it exists, the one below.
  const int bottom_row = (h - 2) / th + 1;
  k_flag_tile_rows<<<(sx * 4 + 255) / 256, 256, 0, s>>>(stamps + cap, sx, bottom_row, pass, halo_flags);      // array 1: written by odd passes
  return hipGetLastError();
}

// Does a transform of this plane, started from its seeds, repair pass 0's seams with bands and strips (relax_pass)?
bool relax_uses_seam_repair(int h, int w, bool seed_bits, int slice_h, bool padded, size_t seam_min_px) {
  const int ax = (w + RX_TW - 1) / RX_TW, ay = (h + RX_NW * RX_P - 1) / (RX_NW * RX_P);
  // (a stack of slices takes it too: slice walls are rows of pinned pixels, wherever they fall in a band or a strip slice)
  (void)slice_h;
  return seed_bits && !padded && (w & 3) == 0 && ax >= 2 && ay >= 2 && (size_t)h * (size_t)w >= seam_min_px;
}

hipError_t relax_pass(hipStream_t s, const uint8_t *img, size_t img_stride, uint32_t *keys, int h, int w,
                      uint32_t max_level, uint32_t pass, uint32_t *stamps, PassFlags pf, uint32_t max_iters,
                      const uint32_t *seed_labels, bool seed_bits, int slice_h, bool carry_checked_later, bool padded,
                      uint32_t *tile_list, size_t seam_min_px, int persistent_pass) {
  const int th = RX_NW * RX_P;
  const int pad = padded ? 1 : 0;
  // A carry out of the 24-bit ring field leaves a finite stamp with ring 0 in the plane (and nothing ever lowers it: the
  // true stamp does not exist).  A transform that hands the finished plane to k_resolve_local lets that kernel look for
  // it, once, instead of every write-back of every pass here (5 VALU ops per pixel in kernels that are VALU-bound).
  const int check_carry = carry_checked_later ? 0 : 1;
  const int sh = slice_h > 0 ? slice_h : h;
  const int ax = (w + RX_TW - 1) / RX_TW, ay = (h + th - 1) / th;     // grid anchored at (0, 0): even passes
  const int sx = ax + 1, sy = ay + 1;                                 // grid shifted by half a tile: odd passes
  // Passes from RX_SAME_GRID_FROM on all run on the anchored grid (relax_todo, read_same): in the long-range regime a tile
  // that stops at its round cap goes on itself, instead of handing its area to the FOUR tiles of the other grid that
  // cover it (each of which loads 8192 pixels to work on a quarter of them).
  static const uint32_t same_from_passes = [] {
    const char *e = tuning_env("WS_RELAX_SAME_GRID_FROM");      // tuning knob, tools/ only
    const uint32_t v = e ? (uint32_t)atoi(e) : RX_SAME_GRID_FROM;
    return v < 3u ? 3u : (v | 1u);                               // odd: the pass before it runs on the anchored grid
  }();
  // The queue in flood order (persistent_pass == 2: the caller has seen sparse seeds, or was told to) starts as early as the
  // schedule allows -- pass 3, right behind the seam repair and one pass with scans: a flood that crosses hundreds of tiles
  // gains six of them from passes 3 .. 6 and pays six launches and their host round trip for it (8192^2, 35 seeds: 0.3 of
  // 4.2 ms).  Pass 4 is then the pass that looks at every tile again; when it finds nothing to change the transform ends
  // inside the replayed graph (run_fused_form: passes 0 .. 4 and the gated resolve).
  static const uint32_t queue_from = [] {
    const char *e = tuning_env("WS_RELAX_QUEUE_FROM");           // tuning knob, tools/ only
    const uint32_t v = e ? (uint32_t)atoi(e) : 3u;
    return v < 3u ? 3u : (v | 1u);
  }();
  const int persist_mode = tuning_env("WS_RELAX_PERSIST") ? atoi(tuning_env("WS_RELAX_PERSIST")) : persistent_pass;      // (A/B knob, tools/ only)
  const bool queue_plane = persist_mode == 2 && tile_list && !pad && (w & 3) == 0 && w >= RX_P &&
                           ((reinterpret_cast<uintptr_t>(img) | img_stride) & 3u) == 0 && relax_tiles(h, w) < (1u << 24);
  // ... and so do the passes themselves on maps of middling seed density (persist_mode 4: the caller has seen between one
  // seed per two tiles and ~30 per tile): same grid from pass 3, scans from pass 2, lists from pass 3.  8192^2 smooth maps,
  // correlation 6 / 8 / 10 / 12 / 16 px: 3.45 / 3.79 / 4.37 / 4.68 / 5.76 -> 3.37 / 3.59 / 4.13 / 4.38 / 5.26 ms; at 4 px
  // (80 seeds per tile) the late schedule wins, 2.96 against 3.12, and a random field never gets that far (gpurun_out/r3ax, r3ay).
  const bool early_queue = (queue_plane || (persist_mode == 4 && tile_list)) && queue_from < same_from_passes;
  const uint32_t same_from = early_queue ? queue_from : same_from_passes;
  const int read_same = pass >= same_from ? 1 : 0, write_same = pass + 1 >= same_from ? 1 : 0;
  const uint32_t list_cap = (uint32_t)relax_tiles(h, w);      // entries per tile list
  const int shifted = read_same ? 0 : (int)(pass & 1u);
  const int tx = shifted ? sx : ax, ty = shifted ? sy : ay;
  const size_t cap = relax_tiles(h, w) * 4;
  const uint32_t *prev = stamps + ((pass + 1) & 1) * cap;
  uint32_t *cur = stamps + (pass & 1) * cap;
  const int ox_ = read_same ? ax : (shifted ? ax : sx), oy_ = read_same ? ay : (shifted ? ay : sy);        // the previous pass's grid
  // Pass 0 only has to produce a good first guess: pass 1 re-examines every pixel on the shifted grid
  // anyway (a capped tile raises all four of its quadrant flags), so its last round -- the one that
  // finds nothing left to do, a third of its time on the bench field -- is not worth running.
  static const uint32_t p0_rounds = [] {
    const char *e = tuning_env("WS_RELAX_P0_ROUNDS");        // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : 2u;
  }();
  static const bool no_seam = tuning_env("WS_RELAX_NO_SEAM") != nullptr;      // A/B knob, tools/ only
  const bool seam_flow = !no_seam && seed_labels != nullptr && relax_uses_seam_repair(h, w, seed_bits, slice_h, padded, seam_min_px);      // (below)
  constexpr uint32_t SEAM_P0_ROUNDS = 6;      // two rounds of four sweeps, then up to four of one (k_relax, chunk == 3)
  const uint32_t p0_cap = seam_flow ? SEAM_P0_ROUNDS : p0_rounds;
  if (pass == 0 && p0_cap < max_iters) max_iters = p0_cap;
  // Late passes (the long-range regime of smooth maps: a few hundred tiles along the flood fronts per pass) end when their
  // SLOWEST tile ends, and a tile that the front is crossing diagonally can take a dozen rounds.  Capping the rounds lets a
  // pass end after the typical tile's work: a capped tile raises all four quadrant flags (like a capped pass-0 tile), so
  // the tiles of the other grid that cover it carry on in the next pass -- next to the front, which has moved on meanwhile.
  static const uint32_t late_cap = [] {
    const char *e = tuning_env("WS_RELAX_LATE_CAP");      // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : RX_LATE_ROUND_CAP;
  }();
  static const uint32_t scan_from_knob = [] {
    const char *e = tuning_env("WS_RELAX_SCAN_FROM");     // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : RX_SCAN_FROM_PASS;
  }();
  const uint32_t scan_from = early_queue ? same_from - 1u : std::max(scan_from_knob, 1u);
  if (pass >= scan_from && late_cap != 0 && late_cap < max_iters) max_iters = late_cap;
  // (tuning knob, tools/ only: "n,c" -- the first n scan passes with c rounds instead)
  if (const char *e = tuning_env("WS_RELAX_WIDE_CAP")) {
    int n_ = 0, c_ = 0;
    if (sscanf(e, "%d,%d", &n_, &c_) == 2 && pass >= scan_from && pass < scan_from + (uint32_t)n_) max_iters = (uint32_t)c_;
  }
  // Passes 1 .. 3 have no scans: on a smooth map a tile that iterates to its own fixpoint by sweeps alone takes up to 64
  // rounds to carry a flood across its 256 columns, all 8192 tiles of them, in a pass that the scan passes then redo.
  static const uint32_t early_cap = [] {
    const char *e = tuning_env("WS_RELAX_EARLY_CAP");     // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : RX_EARLY_ROUND_CAP;
  }();
  if (pass >= 1 && pass < scan_from && early_cap != 0 && early_cap < max_iters) max_iters = early_cap;
  static const uint32_t lite_from = [] {
    const char *e = tuning_env("WS_RELAX_LITE_FROM");     // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : 2u;
  }();
  // passes 0 and 1 run every tile and pass 2 about half of them (bench field): one tile per workgroup
  static const uint32_t chunk_from = [] {
    const char *e = tuning_env("WS_RELAX_CHUNK_FROM");      // tuning knob, tools/ only
    return e ? (uint32_t)atoi(e) : 3u;
  }();
  // Seam repair (k_relax, SEAM): a transform that starts from its seeds runs pass 0 to every tile's own fixpoint and then,
  // as "pass 1", 8-row bands astride the horizontal seams of the 256 x 32 grid and 8-column strips astride the vertical
  // ones -- a third of the pixels of the shifted grid's pass, which it replaces -- and those raise the flags that pass 2
  // reads.  Planes it is not offered for (odd widths, stacks of slices, the virtual halo, planes of a tile or two) keep
  // the alternating grids from pass 1 on.
  // (8192^2 bench field: pass 0 141 -> 168 us, pass 1 117 us -> bands 34 + strips 30 us, the later passes as before: 0.622 ->
  // 0.595 ms per transform; 2048^2: 0.131 -> 0.137 ms, one more launch in a transform that is all launch gaps -- hence the
  // size threshold.  Wider bands, smaller strip slices and a second, shifted strip launch were measured too: no better,
  // profiles/r2_v6_seam_ab.log.)
  // Round caps of that flow: three rounds bring a tile of the bench field to its own fixpoint (the third finds nothing to
  // do); a tile, band or strip slice of a smooth map that is still moving then asks for a re-run in pass 2 instead of
  // carrying a flood across its 256 columns sweep by sweep.
  constexpr uint32_t SEAM_REPAIR_ROUNDS = 4;
  if (seam_flow && pass == 1) {
    // Rows each side of a seam (tuning knob, tools/ only).  What pass 0 leaves wrong thins out fourfold per pixel of distance from
    // the seam, and a band raises a flag when its first or last row changes: with 4 rows a side 41 % of the tiles are flagged
    // (pass 2: 52 us), with 6 a tile in eight (25 us), with 8 one in eleven (23 us) -- and the bands cost 35 / 50 / 59 us.
    static const int seam_band = [] { const char *e = tuning_env("WS_RELAX_SEAM_BAND"); return e ? atoi(e) : 6; }();
    if (seam_band == 4)
      k_relax<2, false, false, false, false, 1><<<ax * (ay - 1), 128, 0, s>>>(img, img_stride, keys, h, w, ax, ay - 1, sx, sy, 0, 1, max_level, pass, prev, cur,
                                                                            pf, SEAM_REPAIR_ROUNDS, nullptr, 0, sh, check_carry, pad, tile_list, 0, 0, 0, list_cap, 0);
    else if (seam_band == 6)
      k_relax<3, false, false, false, false, 1><<<ax * (ay - 1), 192, 0, s>>>(img, img_stride, keys, h, w, ax, ay - 1, sx, sy, 0, 1, max_level, pass, prev, cur,
                                                                            pf, SEAM_REPAIR_ROUNDS, nullptr, 0, sh, check_carry, pad, tile_list, 0, 0, 0, list_cap, 0);
    else
      k_relax<4, false, false, false, false, 1><<<ax * (ay - 1), 256, 0, s>>>(img, img_stride, keys, h, w, ax, ay - 1, sx, sy, 0, 1, max_level, pass, prev, cur,
                                                                            pf, SEAM_REPAIR_ROUNDS, nullptr, 0, sh, check_carry, pad, tile_list, 0, 0, 0, list_cap, 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int strips_x = (ax - 1 + 31) / 32;
    k_relax<RX_NW, false, false, false, false, 2><<<strips_x * ay, 64 * RX_NW, 0, s>>>(img, img_stride, keys, h, w, strips_x, ay, sx, sy, 0, 1, max_level, pass,
                                                                                       prev, cur, pf, SEAM_REPAIR_ROUNDS, nullptr, 0, sh, check_carry, pad, tile_list, 0, 0,
                                                                                       0, list_cap, 0);
    return hipGetLastError();
  }
  const int sb = pass == 0 && seed_labels && seed_bits ? 1 : 0;
  const uint32_t *sl = pass == 0 ? seed_labels : nullptr;
  // (pass 2 after a seam repair runs a tile in ten: a workgroup per four tiles, as in the later passes)
  const uint32_t chunk_from_now = seam_flow ? std::min(chunk_from, 2u) : chunk_from;
  if (pass < chunk_from_now && pass < lite_from) {
    k_relax<RX_NW, false, false, false><<<tx * ty, 64 * RX_NW, 0, s>>>(img, img_stride, keys, h, w, tx, ty, ox_, oy_, shifted, seam_flow && pass == 0 ? 3 : 1, max_level, pass,
                                                                       prev, cur, pf, max_iters, sl, sb, sh, check_carry, pad, tile_list, 0, read_same, write_same, list_cap, 0);
  } else if (pass < chunk_from_now) {
    k_relax<RX_NW, false, false, true><<<tx * ty, 64 * RX_NW, 0, s>>>(img, img_stride, keys, h, w, tx, ty, ox_, oy_, shifted, 1, max_level, pass,
                                                                      prev, cur, pf, max_iters, sl, sb, sh, check_carry, pad, tile_list, 0, read_same, write_same, list_cap, 0);
  } else {
    const int chunk = 4;
    const unsigned grid = (unsigned)((tx * ty + chunk - 1) / chunk);
    static const uint32_t list_from_passes = [] {
      const char *e = tuning_env("WS_RELAX_LIST_FROM");      // tuning knob, tools/ only
      return e ? (uint32_t)atoi(e) : RX_LIST_FROM_PASS;
    }();
    const uint32_t list_from = early_queue ? same_from : list_from_passes;
    if (pass < scan_from) {
      k_relax<RX_NW, true, false, true><<<grid, 64 * RX_NW, 0, s>>>(img, img_stride, keys, h, w, tx, ty, ox_, oy_, shifted, chunk, max_level,
                                                              pass, prev, cur, pf, max_iters, nullptr, 0, sh, check_carry, pad, tile_list, 0, read_same, write_same, list_cap, 0);
    } else if (tile_list && pass >= list_from && pass >= scan_from + 1) {
      // (two passes earlier a launch has cleared this pass's counter: every kernel variant does, given a list)
      // From the second same-grid pass on the list is there already: the tiles of the pass before appended it.
      static const bool no_append = tuning_env("WS_RELAX_NO_APPEND") != nullptr;      // A/B knob, tools/ only
      const uint32_t first_list_pass = std::max(list_from, scan_from + 1);
      const int append_next = !no_append && pass >= same_from && pass >= first_list_pass ? 1 : 0;
      const bool appended = !no_append && pass >= 1 && pass - 1 >= same_from && pass - 1 >= first_list_pass;
      // The same-grid passes run on 128 x 64 tiles (k_relax, SPLIT); the first of them builds its list from the stamps the
      // 256 x 32 grid left behind.
      static const bool no_split = tuning_env("WS_RELAX_NO_SPLIT") != nullptr;      // A/B knob, tools/ only
      const bool split = !no_split && pass >= same_from && same_from >= first_list_pass;
      const int gx = split ? (w + RX_STW - 1) / RX_STW : tx, gy = split ? (h + RX_STH - 1) / RX_STH : ty;
      if (!appended) {
        // (one thread per tile and a few more: the queued marks, dummy slot included, are cleared here)
        const unsigned blocks = (unsigned)((std::max<size_t>((size_t)gx * gy, list_cap + 1) + 255) / 256);
        // The first same-grid pass as ONE persistent launch (k_relax, PERSIST): workgroups pull tiles from a queue and a tile
        // that changes something its neighbour must see queues that neighbour at once.  The pass after it runs every tile
        // from an all-tiles list, so the fixpoint is certified by the ordinary machinery whatever the queue did.
        // 8192^2 smooth maps, correlation length 4 / 16 / 64 / 256 px (profiles/r3_v1_persistent_ab.txt): 3.0 / 5.8 / 6.7 / 3.5 ms
        // with the passes; first come (mode 1) 3.0 / 6.4 / 6.4 / 4.1 -- a tile run costs 13-17 us either way (4 us of loads past
        // L2, 4-8 of scan rounds, 3 of write-through stores, 2 of queue atomics), the queue saves the launch gaps and the
        // tails of the passes and pays for them with a sixth more tile runs (a tile runs on the first flag instead of on all
        // flags of a pass); in flood order (mode 2, from pass 3, on 128 x 128 tiles) 4.2 / 5.8 / 3.9 / 3.0: a third of the
        // tile runs, and a win where floods are long.  The context picks mode 2 by itself when seeds are sparse
        // (run_fused_form); ws_ctx_set_persistent_pass forces or forbids.
        const bool persist = (persist_mode == 1 || persist_mode == 2) && split && pass == same_from && !pad && (w & 3) == 0 && w >= RX_P &&
                             ((reinterpret_cast<uintptr_t>(img) | img_stride) & 3u) == 0 && (size_t)gx * gy <= list_cap && list_cap < (1u << 24);
        if (persist) {
          const bool in_order = persist_mode == 2;      // buckets in flood order (PERSIST == 2) instead of the first-come ring
          hipError_t e = hipMemsetAsync(tile_list + RL_HDR, 0, 2 * (size_t)list_cap * sizeof(uint32_t), s);      // the ring: no entry yet
          if (e == hipSuccess) e = hipMemsetAsync(tile_list + 8, 0, (RL_HDR - 8) * sizeof(uint32_t), s);      // counters (and diagnostics)
          if (e == hipSuccess && in_order)
            e = hipMemsetAsync(tile_list + pq_base(list_cap), 0, (PQ_HDR + (size_t)PQ_B * pq_words_per_bucket(list_cap)) * sizeof(uint32_t), s);
          if (e != hipSuccess) return e;
          const int qy = (h + RX_QTH - 1) / RX_QTH;      // (flood order: the queue's own grid, 128 x 128)
          if (in_order)
            k_relax_list_regrid<RX_STW, RX_QTH, RX_TW, RX_NW * RX_P><<<blocks, 256, 0, s>>>(h, w, gx, qy, ax, ay, pass, prev, tile_list, list_cap, 2);
          else
            k_relax_list_regrid<RX_STW, RX_STH, RX_TW, RX_NW * RX_P><<<blocks, 256, 0, s>>>(h, w, gx, gy, ax, ay, pass, prev, tile_list, list_cap, 1);
          if ((e = hipGetLastError()) != hipSuccess) return e;
          // In flood order: one worker per CU.  Two (all that are resident) run twice as long each, the rounds too -- a
          // workgroup is one wave per SIMD, and two share their vector issue -- so nothing is gained where all are busy, and
          // where most are idle their looks at the counts are in the way: 8192^2 smooth maps, correlation 4 / 16 / 64 / 256 px,
          // 3.89 / 7.75 / 5.73 / 4.14 ms with 512 workers, 3.82 / 7.22 / 4.98 / 3.98 with 256 (gpurun_out/r3z).
          const unsigned workers = tuning_env("WS_RELAX_PERSIST_WORKERS") ? (unsigned)atoi(tuning_env("WS_RELAX_PERSIST_WORKERS"))
                                                                          : std::min<unsigned>(in_order ? RX_LIST_GRID / 2 : RX_LIST_GRID * RX_NW / RX_SNW, (unsigned)(gx * (in_order ? qy : gy)));
          const uint32_t cap = tuning_env("WS_RELAX_PERSIST_CAP") ? (uint32_t)atoi(tuning_env("WS_RELAX_PERSIST_CAP")) : RLQ_ROUND_CAP;
          const int mode = tuning_env("WS_RELAX_PERSIST_MODE") ? atoi(tuning_env("WS_RELAX_PERSIST_MODE")) : 0;
          if (in_order)
            k_relax<RX_QNW, true, true, true, true, 0, 2><<<workers, 64 * RX_QNW, 0, s>>>(img, img_stride, keys, h, w, gx, qy, gx, qy, 0, chunk, max_level, pass, prev, cur, pf, cap,
                                                                                         nullptr, 0, sh, check_carry, pad, tile_list, mode, 1, 1, list_cap, 1);
          else
            k_relax<RX_SNW, true, true, true, true, 0, 1><<<workers, 64 * RX_SNW, 0, s>>>(img, img_stride, keys, h, w, gx, gy, gx, gy, 0, chunk, max_level, pass, prev, cur, pf, cap,
                                                                                         nullptr, 0, sh, check_carry, pad, tile_list, mode, 1, 1, list_cap, 1);
          if ((e = hipGetLastError()) != hipSuccess) return e;
          hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
          (void)hipStreamIsCapturing(s, &capturing);
          if (tuning_env("WS_RELAX_PERSIST_DIAG") && capturing == hipStreamCaptureStatusNone) {      // tools/ only: what the workers did
            uint32_t hd[RL_HDR];
            (void)hipStreamSynchronize(s);
            (void)hipMemcpy(hd, tile_list, sizeof hd, hipMemcpyDeviceToHost);
            fprintf(stderr, "[ws] persistent pass %u: pushed %u popped %u pending %u done %u runs %u | workers wait %.1f us, life %.1f us (sums / 512), polls %u\n", pass,
                    hd[RLQ_TAIL], hd[RLQ_HEAD], hd[RLQ_PENDING], hd[RLQ_DONE], hd[RLQ_RUNS], hd[RLQ_WAIT] / 100.0 / 512.0, hd[RLQ_LIFE] / 100.0 / 512.0, hd[RLQ_POLLS]);
            const double runs = hd[RLQ_RUNS] ? hd[RLQ_RUNS] : 1;
            fprintf(stderr, "[ws]   per tile run (us): load %.2f rounds %.2f epilogue %.2f hand-in %.2f; shader clock %.0f MHz\n", hd[RLQ_PHASE] / 100.0 / runs, hd[RLQ_PHASE + 1] / 100.0 / runs,
                    hd[RLQ_PHASE + 2] / 100.0 / runs, hd[RLQ_PHASE + 3] / 100.0 / runs, hd[RLQ_LIFE] ? 256.0 * hd[RLQ_PHASE + 4] / hd[RLQ_LIFE] * 100.0 : 0.0);
          }
          k_relax_list_all<RX_STW, RX_STH><<<(unsigned)((std::max<size_t>((size_t)gx * gy, list_cap + 1) + 255) / 256), 256, 0, s>>>(h, w, gx, gy, pass + 1, tile_list, list_cap);
          return hipGetLastError();
        }
        if (split && pass == same_from)
          k_relax_list_regrid<RX_STW, RX_STH, RX_TW, RX_NW * RX_P><<<blocks, 256, 0, s>>>(h, w, gx, gy, ax, ay, pass, prev, tile_list, list_cap, 0);
        else if (split)
          k_relax_list<RX_STW, RX_STH><<<blocks, 256, 0, s>>>(h, w, gx, gy, gx, gy, 0, pass, prev, tile_list, 1, list_cap);
        else
          k_relax_list<RX_TW, RX_NW * RX_P><<<blocks, 256, 0, s>>>(h, w, tx, ty, ox_, oy_, shifted, pass, prev, tile_list, read_same, list_cap);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
      }
      if (split)
        k_relax<RX_SNW, true, true, true, true><<<std::min<unsigned>(RX_LIST_GRID * RX_NW / RX_SNW, (unsigned)(gx * gy)), 64 * RX_SNW, 0, s>>>(
            img, img_stride, keys, h, w, gx, gy, gx, gy, 0, chunk, max_level, pass, prev, cur, pf, max_iters, nullptr, 0, sh,
            check_carry, pad, tile_list, 1, 1, 1, list_cap, append_next);
      else
        k_relax<RX_NW, true, true, true><<<std::min<unsigned>(RX_LIST_GRID, (unsigned)(tx * ty)), 64 * RX_NW, 0, s>>>(
            img, img_stride, keys, h, w, tx, ty, ox_, oy_, shifted, chunk, max_level, pass, prev, cur, pf, max_iters, nullptr, 0, sh,
            check_carry, pad, tile_list, 1, read_same, write_same, list_cap, append_next);
    } else {
      k_relax<RX_NW, true, true, true><<<grid, 64 * RX_NW, 0, s>>>(img, img_stride, keys, h, w, tx, ty, ox_, oy_, shifted, chunk, max_level,
                                                             pass, prev, cur, pf, max_iters, nullptr, 0, sh, check_carry, pad, tile_list, 0, read_same, write_same, list_cap, 0);
    }
  }
  return hipGetLastError();
}

}  // namespace wsk
