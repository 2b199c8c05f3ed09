// ws_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the watershed engine.
//
// No MFMA anywhere: the path is integer min/max/compare work on u8 / u32 planes.  The
// rules that matter are coalesced HBM rows, LDS-resident tiles, conflict-free LDS rows
// (lanes run along x) and wave-level reductions for convergence.
#include "ws_common.hpp"

#include <algorithm>
#include <cstdlib>

// per-workgroup phase stamps: only in the diagnostic builds under tools/ (they define the macro)
#ifndef WS_STAMP
#define WS_STAMP(slot) do {} while (0)
#define WS_STAMP_VALUE(slot, v) do {} while (0)
#endif

namespace wsk {

// ---------------------------------------------------------------- small utilities ----

__global__ void k_fill_u32(uint32_t *p, size_t n, uint32_t v) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) p[i] = v;
}

hipError_t fill_u32(hipStream_t s, uint32_t *p, size_t n, uint32_t v) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_fill_u32<<<blocks, 256, 0, s>>>(p, n, v);
  return hipGetLastError();
}

// Zeroes a plane and two small arrays in ONE launch: the prologue of a fused transform clears the
// label plane, the tile-edge stamps and the flag words, and every separate memset is a launch plus
// a dependency gap (~10 us each).
typedef uint32_t u32x4_z __attribute__((ext_vector_type(4)));
__global__ void k_zero3(uint32_t *a, size_t na, uint32_t *b, size_t nb, uint32_t *c, size_t nc) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  size_t head = (size_t)((16u - (unsigned)(reinterpret_cast<uintptr_t>(a) & 15u)) & 15u) >> 2;   // words up to 16-byte alignment
  if (head > na) head = na;
  const size_t body = (na - head) >> 2;
  u32x4_z *a4 = reinterpret_cast<u32x4_z *>(a + head);
  for (size_t i = tid; i < body; i += step) a4[i] = u32x4_z{0u, 0u, 0u, 0u};
  if (tid < head) a[tid] = 0u;
  for (size_t i = head + body * 4 + tid; i < na; i += step) a[i] = 0u;
  for (size_t i = tid; i < nb; i += step) b[i] = 0u;
  for (size_t i = tid; i < nc; i += step) c[i] = 0u;
}

hipError_t zero3(hipStream_t s, uint32_t *a, size_t na, uint32_t *b, size_t nb, uint32_t *c, size_t nc) {
  const size_t most = std::max(std::max(na / 4, nb), std::max(nc, (size_t)1));
  const int blocks = (int)std::min<size_t>((most + 255) / 256, 4096);
  k_zero3<<<blocks, 256, 0, s>>>(a, na, b, nb, c, nc);
  return hipGetLastError();
}

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ void k_random_field(uint8_t *img, size_t stride, int h, int w, uint64_t base) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t n = (size_t)h * w, step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const size_t r = i / (size_t)w, c = i % (size_t)w;
    img[r * stride + c] = (uint8_t)(mix64(base + i) % 254u);
  }
}

hipError_t random_field(hipStream_t s, uint8_t *img, size_t stride, int h, int w, uint64_t seed) {
  const size_t n = (size_t)h * w;
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 16384 ? (n + 1023) / 1024 : 16384);
  k_random_field<<<blocks, 256, 0, s>>>(img, stride, h, w, seed << 40);
  return hipGetLastError();
}

// lib.rs:1670-1677: colour i+1 at seed i; a later duplicate overwrites an earlier one, i.e. the
// LARGEST colour painted on a pixel wins.  One atomicMax per seed costs ~35 ns of memory-side atomic
// each (7.3 M seeds: 250 us), so the painting is split: plain stores first (duplicates race, any of
// them lands), then a fix-up pass in which a seed whose colour is larger than what it finds there
// raises the pixel with atomicMax -- only duplicate seeds ever issue an atomic.
__global__ void k_scatter_seeds(const uint32_t *__restrict__ seeds_rc, const uint32_t *__restrict__ colours, size_t n, int ph,
                                int pw, uint32_t *labels, uint32_t *keys, uint32_t *err_flag, uint32_t *unsorted_flag) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint2 rc = reinterpret_cast<const uint2 *>(seeds_rc)[i];
    if (rc.x >= (uint32_t)ph || rc.y >= (uint32_t)pw) { atomicExch(err_flag, 1u); continue; }
    const size_t p = (size_t)rc.x * pw + rc.y;
    labels[p] = colours ? colours[i] : (uint32_t)(i + 1);
    if (keys) keys[p] = 0u;
    // a strictly increasing (row-major) list -- what find_local_minima returns -- cannot hold duplicates;
    // anything else arms the fix-up pass (plain idempotent store)
    if (i + 1 < n) {
      const uint2 nx = reinterpret_cast<const uint2 *>(seeds_rc)[i + 1];
      if ((size_t)nx.x * pw + nx.y <= p) *unsorted_flag = 1u;
    }
  }
}

__global__ void k_scatter_fixup(const uint32_t *__restrict__ seeds_rc, const uint32_t *__restrict__ colours, size_t n, int ph,
                                int pw, uint32_t *labels, const uint32_t *unsorted_flag) {
  if (*unsorted_flag == 0u) return;          // no duplicates possible: nothing to fix
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint2 rc = reinterpret_cast<const uint2 *>(seeds_rc)[i];
    if (rc.x >= (uint32_t)ph || rc.y >= (uint32_t)pw) continue;
    const size_t p = (size_t)rc.x * pw + rc.y;
    const uint32_t mine = colours ? colours[i] : (uint32_t)(i + 1);
    if (labels[p] < mine) atomicMax(&labels[p], mine);
  }
}

// err_flag points at two consecutive words: [0] seed out of bounds, [1] seed list not strictly increasing
hipError_t scatter_seeds(hipStream_t s, const uint32_t *seeds_rc, const uint32_t *colours, size_t n, int ph, int pw,
                         uint32_t *labels, uint32_t *keys, uint32_t *err_flag) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  k_scatter_seeds<<<blocks, 256, 0, s>>>(seeds_rc, colours, n, ph, pw, labels, keys, err_flag, err_flag + 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  k_scatter_fixup<<<blocks, 256, 0, s>>>(seeds_rc, colours, n, ph, pw, labels, err_flag + 1);
  return hipGetLastError();
}

// ---- painting a whole label plane in one pass ------------------------------------------------
//
// memset + scatter writes the plane twice (the scatter's 4-byte stores dirty nearly every line of it
// again).  When the seed list is sorted by pixel index -- what find_local_minima returns -- a wave can
// instead FIND the seeds of its own 4096-pixel chunk (64-ary lower bound: 4 dependent probes for
// 7 M seeds) and walk down the list from there: per 256-pixel segment the seeds' colours are dropped
// into a 1 KiB LDS row and the segment is written once, 16 B per lane.
// Nothing is assumed: every wave also checks a slice of the list for order (and every seed for
// bounds); if any consecutive pair is out of order `unsorted` is raised and k_scatter_fixup, which
// runs next, raises every seed pixel to its largest seed index with atomicMax.  That is exact
// whatever the wave walks found, because a painted pixel always holds 0 or the colour of one of its
// own seeds.
constexpr int PAINT_SEG = 256;       // pixels per step of a wave: one 16-byte store per lane
constexpr int PAINT_STEPS = 16;      // consecutive segments per wave: one search, then a walk down the list
constexpr int PAINT_WAVES = 4;       // waves per workgroup

__device__ __forceinline__ unsigned long long seed_pos(uint2 rc, int pw) { return (unsigned long long)rc.x * (unsigned)pw + rc.y; }

// order and bounds of one wave's slice of the list.  flags: [0] seed out of bounds, [1] list not
// sorted, [2] list not strictly increasing
__device__ __forceinline__ void check_seed_slice(const uint2 *__restrict__ seeds, size_t n, int ph, int pw, size_t slice,
                                                 size_t nslices, int lane, uint32_t *flags) {
  const size_t per = (n + nslices - 1) / nslices;
  const size_t lo = slice * per, hi = lo + per < n ? lo + per : n;
  for (size_t j = lo + lane; j < hi; j += 64) {
    const uint2 a = seeds[j];
    if (a.x >= (uint32_t)ph || a.y >= (uint32_t)pw) atomicExch(&flags[0], 1u);
    if (j + 1 < n) {
      const unsigned long long pa = seed_pos(a, pw), pb = seed_pos(seeds[j + 1], pw);
      if (pb < pa) flags[1] = 1u;
      if (pb <= pa) flags[2] = 1u;
    }
  }
}

// 64-ary lower bound, one probe per lane and round: smallest j with pos(j) >= p0 (4 rounds for 7 M seeds).
// Terminates and stays inside [0, n] whatever the order of the list.
__device__ __forceinline__ size_t seed_lower_bound(const uint2 *__restrict__ seeds, size_t n, int pw, unsigned long long p0, int lane) {
  size_t lo = 0, hi = n;
  while (hi - lo > 64) {
    const size_t step = (hi - lo + 63) / 64;
    const size_t j = lo + (size_t)lane * step;
    const bool before = j < hi && seed_pos(seeds[j], pw) < p0;
    const int c = __popcll(__builtin_amdgcn_ballot_w64(before));
    if (c == 0) { hi = lo; break; }
    const size_t nlo = lo + (size_t)(c - 1) * step + 1, nhi = lo + (size_t)c * step;
    hi = nhi < hi ? nhi : hi;
    lo = nlo;
  }
  if (hi > lo) {
    const size_t j = lo + lane;
    const bool before = j < hi && seed_pos(seeds[j], pw) < p0;
    lo += __popcll(__builtin_amdgcn_ballot_w64(before));
  }
  return lo;
}

// Two lower bounds (pa <= pb) in one walk: the probes of both searches are in flight together, so a wave that needs both
// ends of its chunk pays the dependent round trips of one search (k_seed_tables is little else: 8 of its ~12).
__device__ __forceinline__ void seed_lower_bound2(const uint2 *__restrict__ seeds, size_t n, int pw, unsigned long long pa,
                                                  unsigned long long pb, int lane, size_t &out_a, size_t &out_b) {
  if (n == 0) { out_a = out_b = 0; return; }      // (no list to probe)
  size_t lo[2] = {0, 0}, hi[2] = {n, n};
  const unsigned long long p[2] = {pa, pb};
  while (hi[0] - lo[0] > 64 || hi[1] - lo[1] > 64) {
    size_t step[2], j[2];
    uint2 probe[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      step[k] = (hi[k] - lo[k] + 63) / 64;
      j[k] = lo[k] + (size_t)lane * step[k];
      probe[k] = seeds[j[k] < hi[k] ? j[k] : n - 1];
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (hi[k] - lo[k] <= 64) continue;      // uniform
      const bool before = j[k] < hi[k] && seed_pos(probe[k], pw) < p[k];
      const int c = __popcll(__builtin_amdgcn_ballot_w64(before));
      if (c == 0) { hi[k] = lo[k]; continue; }
      const size_t nlo = lo[k] + (size_t)(c - 1) * step[k] + 1, nhi = lo[k] + (size_t)c * step[k];
      hi[k] = nhi < hi[k] ? nhi : hi[k];
      lo[k] = nlo;
    }
  }
  uint2 last[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) last[k] = seeds[lo[k] + lane < hi[k] ? lo[k] + lane : n - 1];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const bool before = hi[k] > lo[k] && lo[k] + lane < hi[k] && seed_pos(last[k], pw) < p[k];
    lo[k] += __popcll(__builtin_amdgcn_ballot_w64(before));
  }
  out_a = lo[0];
  out_b = lo[1];
}

__global__ __launch_bounds__(64 * PAINT_WAVES) void k_paint_sorted(const uint32_t *__restrict__ seeds_rc, size_t n, int ph, int pw,
                                                                  uint32_t *labels, size_t npx, size_t nchunk, int steps,
                                                                  uint32_t *err_flag,
                                                                  uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b) {
  __shared__ __attribute__((aligned(16))) uint32_t sRow[PAINT_WAVES][PAINT_SEG];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2 *seeds = reinterpret_cast<const uint2 *>(seeds_rc);
  {   // side job: the small arrays the transform wants zeroed (tile-edge stamps, flag words)
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n_zero_a; i += step) zero_a[i] = 0u;
    for (size_t i = tid; i < n_zero_b; i += step) zero_b[i] = 0u;
  }
  const size_t chunk = (size_t)blockIdx.x * PAINT_WAVES + wave;       // `steps` consecutive segments
  if (chunk >= nchunk) return;
  uint32_t *row = sRow[wave];
  const bool vec_ok = (reinterpret_cast<uintptr_t>(labels) & 15u) == 0;
  check_seed_slice(seeds, n, ph, pw, chunk, nchunk, lane, err_flag);      // independent loads, issued first
  unsigned long long p0 = (unsigned long long)chunk * (unsigned long long)(PAINT_SEG * steps);
  const size_t lo = seed_lower_bound(seeds, n, pw, p0, lane);

  // ---- walk: the list from `lo` on is cut into fixed windows of 64 seeds, one per lane; `off` is the
  // first lane of the current window that has not been painted yet.  A segment paints the run of
  // in-segment seeds from `off` on, across as many windows as it takes.  The window after the current
  // one is always in flight already, so no load latency is exposed after the first.
  size_t wbase = lo;                                   // list index of lane 0 of the current window
  int off = 0;
  uint2 cur = make_uint2(0u, 0u), nxt = make_uint2(0u, 0u);
  if (n) {
    cur = seeds[wbase + lane < n ? wbase + lane : n - 1];
    nxt = seeds[wbase + 64 + lane < n ? wbase + 64 + lane : n - 1];
  }
  for (int sgm = 0; sgm < steps && p0 < npx; ++sgm, p0 += PAINT_SEG) {
    const unsigned long long p1 = p0 + PAINT_SEG < npx ? p0 + PAINT_SEG : npx;
    *reinterpret_cast<u32x4_z *>(&row[lane * 4]) = u32x4_z{0u, 0u, 0u, 0u};
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the zeroed row before the colours
    for (;;) {
      const size_t j = wbase + lane;
      const unsigned long long pos = seed_pos(cur, pw);
      const bool ok = j < n && pos >= p0 && pos < p1;
      // lanes below `off` count as done; the run ends at the first lane from `off` on that is not in the segment
      const unsigned long long m = __builtin_amdgcn_ballot_w64(ok) | ((1ull << off) - 1ull);
      const int end = m == ~0ull ? 64 : __builtin_ctzll(~m);
      if (lane >= off && lane < end && cur.x < (uint32_t)ph && cur.y < (uint32_t)pw)
        atomicMax(&row[(uint32_t)(pos - p0)], (uint32_t)(j + 1));
      off = end;
      if (off < 64) break;
      wbase += 64;                                       // window exhausted: the prefetched one becomes current
      off = 0;
      cur = nxt;
      if (n) nxt = seeds[wbase + 64 + lane < n ? wbase + 64 + lane : n - 1];
      if (wbase >= n) break;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    // one 16-byte store per lane
    const u32x4_z v = *reinterpret_cast<const u32x4_z *>(&row[lane * 4]);
    const unsigned long long q = p0 + (unsigned long long)lane * 4;
    if (q + 4 <= npx && vec_ok) {
      *reinterpret_cast<u32x4_z *>(labels + q) = v;
    } else {
      const uint32_t e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) if (q + k < npx) labels[q + k] = e[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the row is read before it is zeroed again
  }
}

// The side tables of a STRICTLY increasing seed list: one bit per pixel ("is a seed") and, per
// 32-pixel word, the list index of the first seed of the word.  The colour of seed pixel p is then
//     word_base[p / 32] + popcount(mask[p / 32] & ((1 << p % 32) - 1)) + 1,
// 16 MiB of tables instead of a painted 256 MiB plane at 8192^2.  A wave owns TAB_WORDS words
// (8192 pixels): lower bound of the chunk start, then every window of 64 seeds drops its bits into a
// 1 KiB LDS row (no per-segment ordering: windows stream through four at a time), one wave scan of
// the word popcounts gives the bases.  Nothing here repairs a list that is not strictly increasing:
// flags[2] tells the caller, who repeats the transform with paint_labels.
constexpr int TAB_WORDS = 256;       // mask words per wave (8192 pixels): four per lane
constexpr int TAB_WIN = 4;           // windows of 64 list entries in flight per round of the walk

__global__ __launch_bounds__(64 * PAINT_WAVES) void k_seed_tables(const uint32_t *__restrict__ seeds_rc, size_t n, int ph, int pw,
                                                                 uint32_t *mask, uint32_t *word_base, size_t npx, size_t nchunk,
                                                                 uint32_t *flags,
                                                                 uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b,
                                                                 const uint32_t *__restrict__ slice_first, size_t slice_px, uint32_t colour_bias) {
  __shared__ __attribute__((aligned(16))) uint32_t sRow[PAINT_WAVES][TAB_WORDS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint2 *seeds = reinterpret_cast<const uint2 *>(seeds_rc);
  {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n_zero_a; i += step) zero_a[i] = 0u;
    for (size_t i = tid; i < n_zero_b; i += step) zero_b[i] = 0u;
  }
  const size_t chunk = (size_t)blockIdx.x * PAINT_WAVES + wave;
  if (chunk >= nchunk) return;
  uint32_t *row = sRow[wave];
  *reinterpret_cast<u32x4_z *>(&row[lane * 4]) = u32x4_z{0u, 0u, 0u, 0u};
  const unsigned long long p0 = (unsigned long long)chunk * (TAB_WORDS * 32);
  const unsigned long long p1 = p0 + TAB_WORDS * 32 < npx ? p0 + TAB_WORDS * 32 : npx;
  // The list is read ONCE, and the walk is also the proof that it is strictly increasing and in bounds: the wave takes
  // the list range [lo, hi) between the lower bounds of its chunk's two ends and checks that every seed in it lies in
  // the chunk, in the plane, and after its predecessor.  Neighbouring waves compute the bound they share by the same
  // search on the same list, so the ranges tile [0, n) whatever the list looks like (the last wave takes what is left);
  // if every range passes, the whole list is strictly increasing.  Anything else raises flags[2] (and flags[0] for a
  // seed outside the plane) and the caller repeats the transform with paint_labels, whose own checks name the fault.
  // (A separate pass over the list for these checks was 58 of the kernel's 161 MB.)
  size_t lo, hi;
  seed_lower_bound2(seeds, n, pw, p0, p1, lane, lo, hi);
  if (chunk + 1 == nchunk) hi = n;
  bool bad = hi < lo || hi - lo > (size_t)TAB_WORDS * 32;      // more seeds than pixels: not worth walking
  if (bad) hi = lo;
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the zeroed row before the bits

  // TAB_WIN windows of 64 seeds per round, all their loads in flight together (sixteen windows a round -- a chunk of the
  // bench field holds ~890 seeds -- changed nothing: of the kernel's 30 us the two searches are 9, the rest is the list
  // streaming through; profiles/r2_v7_chase_ab.log)
  unsigned long long prev_last = 0;      // position of the seed before the round's first (none: the first of the range)
  bool have_prev = false;
  for (size_t wbase = lo; wbase < hi; wbase += 64 * TAB_WIN) {
    uint2 win[TAB_WIN];
#pragma unroll
    for (int k = 0; k < TAB_WIN; ++k) {
      const size_t j = wbase + 64 * k + lane;
      win[k] = seeds[j < hi ? j : hi - 1];
    }
#pragma unroll
    for (int k = 0; k < TAB_WIN; ++k) {
      const size_t j = wbase + 64 * k + lane;
      const unsigned long long pos = seed_pos(win[k], pw);
      const bool mine = j < hi;
      const bool inside = win[k].x < (uint32_t)ph && win[k].y < (uint32_t)pw;
      // predecessor: the lane below, or the last lane of the window before (a clamped lane repeats the last seed: ignored)
      unsigned long long before = __shfl_up(pos, 1, 64);
      if (lane == 0) before = prev_last;
      const bool first_of_range = lane == 0 && !have_prev;
      if (mine && !inside) atomicExch(&flags[0], 1u);
      if (mine && (!inside || pos < p0 || pos >= p1 || (!first_of_range && pos <= before))) bad = true;
      if (mine && inside && pos >= p0 && pos < p1) {
        const uint32_t b = (uint32_t)(pos - p0);
        atomicOr(&row[b >> 5], 1u << (b & 31u));
      }
      prev_last = __shfl(pos, 63, 64);
      have_prev = true;
    }
  }
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && lane == 0) { flags[1] = 1u; flags[2] = 1u; }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");

  // lane l owns words 4l .. 4l+3 of the chunk: exclusive scan of their popcounts = seeds before them
  const u32x4_z w = *reinterpret_cast<const u32x4_z *>(&row[lane * 4]);
  const uint32_t c0 = __popc(w.x), c1 = __popc(w.y), c2 = __popc(w.z), cnt = c0 + c1 + c2 + __popc(w.w);
  const uint32_t incl = wave_inclusive_sum(cnt);
  uint32_t b0 = (uint32_t)lo + incl - cnt;
  // a stack of slices (slice_px % 128 == 0: a lane's four words lie in one slice): list indices count from the slice's first seed
  if (slice_first) {      // (clamped twice over for lists that are not what they should be: the slice index and the difference)
    const size_t npx_slices = (size_t)((npx + slice_px - 1) / slice_px);
    const uint32_t sf = slice_first[min((size_t)((p0 + 128ull * (unsigned)lane) / slice_px), npx_slices)];
    b0 = b0 >= sf ? b0 - sf : 0u;
  }
  // a row block of a larger field: its seeds are entries [g0, g0 + n) of the caller's list, so colours start at g0 + 1
  b0 += colour_bias;
  const u32x4_z bases = u32x4_z{b0, b0 + c0, b0 + c0 + c1, b0 + c0 + c1 + c2};
  const size_t wi = (size_t)(p0 >> 5) + 4 * lane, nwords = (npx + 31) / 32;
  if (wi + 4 <= nwords && ((reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(word_base)) & 15u) == 0) {
    *reinterpret_cast<u32x4_z *>(mask + wi) = w;
    *reinterpret_cast<u32x4_z *>(word_base + wi) = bases;
  } else {
    const uint32_t wv[4] = {w.x, w.y, w.z, w.w}, bv[4] = {bases.x, bases.y, bases.z, bases.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) if (wi + k < nwords) { mask[wi + k] = wv[k]; word_base[wi + k] = bv[k]; }
  }
}

static int paint_steps() {
  static const int steps = [] {
    const char *e = tuning_env("WS_PAINT_STEPS");      // tuning knob, tools/ only
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : PAINT_STEPS;
  }();
  return steps;
}

// Paints colour i + 1 at seed i over a zeroed plane, later duplicates winning (lib.rs:1672-1677), in
// one pass over the plane plus a fix-up that only works when the list is not sorted.  Also zeroes two
// small arrays.  err_flag points at three consecutive words (out of bounds, unsorted, not strictly
// increasing); they must be zero.
hipError_t paint_labels(hipStream_t s, const uint32_t *seeds_rc, size_t n, int ph, int pw, uint32_t *labels,
                        uint32_t *err_flag, uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b) {
  const size_t npx = (size_t)ph * pw;
  const int steps = paint_steps();
  const size_t per_wave = (size_t)PAINT_SEG * steps;
  const size_t nchunk = (npx + per_wave - 1) / per_wave;
  const size_t blocks = std::max<size_t>((nchunk + PAINT_WAVES - 1) / PAINT_WAVES, 1);
  k_paint_sorted<<<(unsigned)blocks, 64 * PAINT_WAVES, 0, s>>>(seeds_rc, n, ph, pw, labels, npx, nchunk, steps, err_flag,
                                                              zero_a, n_zero_a, zero_b, n_zero_b);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || n == 0) return e;
  const int fb = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
  k_scatter_fixup<<<fb, 256, 0, s>>>(seeds_rc, nullptr, n, ph, pw, labels, err_flag + 1);
  return hipGetLastError();
}

// `mask` and `word_base` hold (ph * pw + 31) / 32 words each.  The caller reads err_flag[2] ("not strictly
// increasing") back and, if it is raised, repeats the transform with paint_labels.
hipError_t seed_tables(hipStream_t s, const uint32_t *seeds_rc, size_t n, int ph, int pw, uint32_t *mask, uint32_t *word_base,
                       uint32_t *err_flag, uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b,
                       const uint32_t *slice_first, size_t slice_px, uint32_t colour_bias) {
  const size_t npx = (size_t)ph * pw;
  const size_t per_wave = (size_t)TAB_WORDS * 32;
  const size_t nchunk = (npx + per_wave - 1) / per_wave;
  const size_t blocks = std::max<size_t>((nchunk + PAINT_WAVES - 1) / PAINT_WAVES, 1);
  k_seed_tables<<<(unsigned)blocks, 64 * PAINT_WAVES, 0, s>>>(seeds_rc, n, ph, pw, mask, word_base, npx, nchunk, err_flag,
                                                             zero_a, n_zero_a, zero_b, n_zero_b, slice_first, slice_px, colour_bias);
  return hipGetLastError();
}

// Seed i of a batch belongs to the slice k with slice_first[k] <= i < slice_first[k + 1]; in the stacked plane it sits
// k * slice_h rows further down.  A seed outside its own slice must not land in a neighbour: it becomes (~0, ~0), which
// every seed kernel reports as out of bounds.
__global__ void k_stack_seeds(const uint2 *__restrict__ seeds, size_t n, const uint32_t *__restrict__ slice_first, size_t n_slices,
                              int slice_h, int pw, uint2 *out, uint32_t shift) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    size_t lo = 0, hi = n_slices;            // largest k with slice_first[k] <= i
    while (hi - lo > 1) {
      const size_t mid = (lo + hi) / 2;
      if (slice_first[mid] <= i) lo = mid; else hi = mid;
    }
    uint2 rc = seeds[i];
    // shift: seed_shift with edge correction (ws_hip.h); compared before the add, so that ~0 cannot wrap into the plane
    const bool ok = rc.x < (uint32_t)slice_h - shift && rc.y < (uint32_t)pw - shift;
    rc.x += shift; rc.y += shift;
    out[i] = ok ? make_uint2(rc.x + (uint32_t)lo * (uint32_t)slice_h, rc.y) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  }
}

hipError_t stack_seeds(hipStream_t s, const uint32_t *seeds_rc, size_t n, const uint32_t *slice_first, size_t n_slices,
                       int slice_h, int pw, uint32_t *stacked_rc, uint32_t shift) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  k_stack_seeds<<<blocks, 256, 0, s>>>(reinterpret_cast<const uint2 *>(seeds_rc), n, slice_first, n_slices, slice_h, pw,
                                       reinterpret_cast<uint2 *>(stacked_rc), shift);
  return hipGetLastError();
}

// seed_shift (ws_hip.h): every seed moves by (+shift, +shift); a coordinate that cannot move without wrapping becomes ~0,
// which every seed kernel reports as out of bounds
__global__ void k_shift_seeds(const uint2 *__restrict__ src, size_t n, uint32_t shift, uint2 *dst) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint2 rc = src[i];
    const bool ok = rc.x < 0xFFFFFFFFu - shift && rc.y < 0xFFFFFFFFu - shift;
    dst[i] = ok ? make_uint2(rc.x + shift, rc.y + shift) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  }
}

hipError_t shift_seeds(hipStream_t s, const uint32_t *src, size_t n, uint32_t shift, uint32_t *dst) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  k_shift_seeds<<<blocks, 256, 0, s>>>(reinterpret_cast<const uint2 *>(src), n, shift, reinterpret_cast<uint2 *>(dst));
  return hipGetLastError();
}

// The host ABI's seeds are pairs of 64-bit words (Rust's (usize, usize)); the engine's are 32-bit.  A coordinate outside
// the (padded) plane -- the reference panics there, lib.rs:1675-1677 -- becomes ~0, which every seed kernel reports
// as out of bounds.  (This loop used to run on the host: 5 ms for the 7.3 M seeds of the bench field.)
__global__ void k_narrow_seeds(const uint64_t *__restrict__ src, size_t n, uint64_t ph, uint64_t pw, uint2 *dst, uint64_t shift) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint64_t r = src[2 * i], c = src[2 * i + 1];      // (ph, pw >= shift: the plane is padded whenever shift is 1)
    dst[i] = (r < ph - shift && c < pw - shift) ? make_uint2((uint32_t)(r + shift), (uint32_t)(c + shift)) : make_uint2(0xFFFFFFFFu, 0xFFFFFFFFu);
  }
}

hipError_t narrow_seeds(hipStream_t s, const uint64_t *src, size_t n, size_t ph, size_t pw, uint32_t *dst, uint32_t shift) {
  if (n == 0) return hipSuccess;
  const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 8192);
  k_narrow_seeds<<<blocks, 256, 0, s>>>(src, n, ph, pw, reinterpret_cast<uint2 *>(dst), shift);
  return hipGetLastError();
}

__global__ void k_widen(const uint32_t *src, uint64_t *dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) dst[i] = src[i];
}

hipError_t widen_labels(hipStream_t s, const uint32_t *src, uint64_t *dst, size_t n) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_widen<<<blocks, 256, 0, s>>>(src, dst, n);
  return hipGetLastError();
}

hipError_t widen_pairs(hipStream_t s, const uint32_t *src, uint64_t *dst, size_t n_values) {
  return widen_labels(s, src, dst, n_values);
}

// ... and back: 64-bit words that are known to fit 32 bits (lake records: a colour and an area of a plane below 2^31 pixels), so
// that they cross the bus as half the bytes and are widened again by the host's threads (ws_hostcopy.hip)
__global__ void k_narrow_words(const uint64_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) dst[i] = (uint32_t)src[i];
}

hipError_t narrow_words(hipStream_t s, const uint64_t *src, uint32_t *dst, size_t n) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_narrow_words<<<blocks, 256, 0, s>>>(src, dst, n);
  return hipGetLastError();
}

// Label plane as the reference's hook sees it after `level` (lib.rs:1796-1804): a pixel
// is coloured once its arrival level is <= level; segmenting colours never change later.
__global__ void k_snapshot(const uint32_t *keys, const uint32_t *labels, uint64_t *dst, size_t n,
                           uint32_t level) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t k = keys[i];
    dst[i] = (k != KEY_INF && (k >> 24) <= level) ? (uint64_t)labels[i] : 0ull;
  }
}

// the same into a u32 plane on the device (ws_level_snapshot_device): 16-byte accesses where the planes allow
__global__ void k_snapshot_u32(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels, uint32_t *dst, size_t n, uint32_t level) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t k = keys[i];
    dst[i] = (k != KEY_INF && (k >> 24) <= level) ? labels[i] : 0u;
  }
}

hipError_t snapshot_level_u32(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *dst, size_t n, uint32_t level) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 16384 ? (n + 1023) / 1024 : 16384);
  k_snapshot_u32<<<blocks, 256, 0, s>>>(keys, labels, dst, n, level);
  return hipGetLastError();
}

hipError_t snapshot_level(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint64_t *dst,
                          size_t n, uint32_t level) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_snapshot<<<blocks, 256, 0, s>>>(keys, labels, dst, n, level);
  return hipGetLastError();
}

// ------------------------------------------------------- tile scheduling helper ----
//
// A tile re-runs in pass p only if a neighbour changed the edge it shares with it in pass p-1
// (edge stamps, double buffered by pass parity; layout per tile: 0 top, 1 bottom, 2 left, 3 right;
// value = pass + 1 of the writer).

__device__ __forceinline__ bool tile_must_run(const uint32_t *stamps_prev, int tx, int ty, int tilesX,
                                              int tilesY, uint32_t pass) {
  if (pass == 0) return true;
  // stamp layout per tile: 0 top, 1 bottom, 2 left, 3 right; value = pass+1 of the writer
  const int t = ty * tilesX + tx;
  bool run = false;
  if (ty > 0) run |= stamps_prev[(size_t)(t - tilesX) * 4 + 1] == pass;
  if (ty + 1 < tilesY) run |= stamps_prev[(size_t)(t + tilesX) * 4 + 0] == pass;
  if (tx > 0) run |= stamps_prev[(size_t)(t - 1) * 4 + 3] == pass;
  if (tx + 1 < tilesX) run |= stamps_prev[(size_t)(t + 1) * 4 + 2] == pass;
  return run;
}

size_t resolve_tiles(int h, int w) { return (size_t)tiles_of(w) * tiles_of(h); }

// ---------------------------------------------------- fused engine: label resolve ----
//
// With the stamps final, the colour of a pixel is the colour of its parent: the first
// neighbour in down,right,left,up order (lib.rs:190) that was coloured strictly earlier
// (lib.rs:237-248, col0).  Parents form a forest rooted at the seeds.  A tile computes the
// parent direction of its 16 pixels per thread once, then pulls labels along the forest
// until nothing changes; labels only ever go 0 -> final, so a coloured pixel is done.

__global__ __launch_bounds__(NTHREADS) void k_resolve(const uint32_t *__restrict__ keys, uint32_t *labels,
                                                      int H, int W, int tilesX, int tilesY, uint32_t pass,
                                                      uint32_t *stamps, PassFlags pf) {
  __shared__ uint32_t sB[LP][LP];
  __shared__ uint32_t s_edges;

  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const size_t ntiles = (size_t)tilesX * tilesY;
  const uint32_t *stamps_prev = stamps + ((pass + 1) & 1) * ntiles * 4;
  uint32_t *stamps_cur = stamps + (pass & 1) * ntiles * 4;
  if (blockIdx.x == 0 && threadIdx.x < NSTRIPE)
    pf.edge_changed[((pass + 1) % COUNTER_RING) * FLAG_SLOT + threadIdx.x * STRIPE_STRIDE] = 0;
  if (!tile_must_run(stamps_prev, tile_x, tile_y, tilesX, tilesY, pass)) return;

  const int tid = threadIdx.x;
  const int lane = tid & 63, strip = tid >> 6;
  const int x0 = tile_x * TS, y0 = tile_y * TS;
  const int lx = lane + 1, ly0 = strip * STRIP + 1;
  const int gx = x0 + lane, gy0 = y0 + strip * STRIP;
  if (tid == 0) s_edges = 0;

  // phase 1: stamps -> LDS, parent directions -> registers
  for (int idx = tid; idx < LP * LP; idx += NTHREADS) {
    const int ly = idx / LP, lxx = idx - ly * LP;
    const int gy = y0 - 1 + ly, gxx = x0 - 1 + lxx;
    uint32_t v = KEY_INF;
    if (gy >= 0 && gy < H && gxx >= 0 && gxx < W) v = keys[(size_t)gy * W + gxx];
    sB[ly][lxx] = v;
  }
  __syncthreads();
  uint32_t dirs = 0;      // 2 bits per pixel: 0 down, 1 right, 2 left, 3 up
  uint32_t has_parent = 0;
#pragma unroll
  for (int i = 0; i < STRIP; ++i) {
    const uint32_t k = sB[ly0 + i][lx];
    const int gy = gy0 + i;
    // flooded pixels are interior pixels (lib.rs:220-222).  On a row block of a larger field the
    // first/last local rows are halo copies whose down/up neighbour is not here, so their parent
    // cannot be decided locally: they are left to the owning rank.
    if (k != 0u && k != KEY_INF && gy >= 1 && gy < H - 1 && gx >= 1 && gx < W - 1) {
      const uint32_t d = sB[ly0 + i + 1][lx], r = sB[ly0 + i][lx + 1];
      const uint32_t l = sB[ly0 + i][lx - 1];
      const uint32_t dir = d < k ? 0u : (r < k ? 1u : (l < k ? 2u : 3u));   // at a fixpoint one of the four is < k
      dirs |= dir << (2 * i);
      has_parent |= 1u << i;
    }
  }
  __syncthreads();

  // phase 2: labels -> the same LDS tile
  for (int idx = tid; idx < LP * LP; idx += NTHREADS) {
    const int ly = idx / LP, lxx = idx - ly * LP;
    const int gy = y0 - 1 + ly, gxx = x0 - 1 + lxx;
    uint32_t v = 0;
    if (gy >= 0 && gy < H && gxx >= 0 && gxx < W) v = labels[(size_t)gy * W + gxx];
    sB[ly][lxx] = v;
  }
  __syncthreads();
  uint32_t own[STRIP];
#pragma unroll
  for (int i = 0; i < STRIP; ++i) own[i] = sB[ly0 + i][lx];

  uint32_t todo = 0;      // pixels with a parent and no colour yet
#pragma unroll
  for (int i = 0; i < STRIP; ++i) todo |= (((has_parent >> i) & 1u) && own[i] == 0u) ? (1u << i) : 0u;
  uint32_t changed_mask = 0;

  for (;;) {
    int changed = 0;
    if (todo) {
#pragma unroll
      for (int i = 0; i < STRIP; ++i) {
        if ((todo >> i) & 1u) {
          const uint32_t dir = (dirs >> (2 * i)) & 3u;
          const uint32_t vd = (i == STRIP - 1) ? sB[ly0 + STRIP][lx] : own[i + 1];
          const uint32_t vu = (i == 0) ? sB[ly0 - 1][lx] : own[i - 1];
          const uint32_t vr = sB[ly0 + i][lx + 1], vl = sB[ly0 + i][lx - 1];
          const uint32_t src = dir == 0u ? vd : (dir == 1u ? vr : (dir == 2u ? vl : vu));
          if (src) { own[i] = src; sB[ly0 + i][lx] = src; todo &= ~(1u << i); changed = 1; changed_mask |= 1u << i; }
        }
      }
#pragma unroll
      for (int i = STRIP - 1; i >= 0; --i) {
        if ((todo >> i) & 1u) {
          const uint32_t dir = (dirs >> (2 * i)) & 3u;
          const uint32_t vd = (i == STRIP - 1) ? sB[ly0 + STRIP][lx] : own[i + 1];
          const uint32_t vu = (i == 0) ? sB[ly0 - 1][lx] : own[i - 1];
          const uint32_t vr = sB[ly0 + i][lx + 1], vl = sB[ly0 + i][lx - 1];
          const uint32_t src = dir == 0u ? vd : (dir == 1u ? vr : (dir == 2u ? vl : vu));
          if (src) { own[i] = src; sB[ly0 + i][lx] = src; todo &= ~(1u << i); changed = 1; changed_mask |= 1u << i; }
        }
      }
    }
    if (!__syncthreads_or(changed)) break;
  }

  if (gx < W) {
#pragma unroll
    for (int i = 0; i < STRIP; ++i) {
      const int gy = gy0 + i;
      if (((changed_mask >> i) & 1u) && gy < H) labels[(size_t)gy * W + gx] = own[i];
    }
  }
  uint32_t e = 0;
  if (strip == 0 && (changed_mask & 1u)) e |= 1u;
  if (strip == (NTHREADS / 64) - 1 && (changed_mask >> (STRIP - 1))) e |= 2u;
  if (lane == 0 && changed_mask) e |= 4u;
  if (lane == 63 && changed_mask) e |= 8u;
  if (changed_mask) e |= 16u;                                             // anything at all
  if (e) atomicOr(&s_edges, e);
  __syncthreads();
  if (tid == 0) {
    const uint32_t ed = s_edges;
    if (ed) {
      const size_t t = (size_t)tile_y * tilesX + tile_x;
      if (ed & 1u) stamps_cur[t * 4 + 0] = pass + 1;
      if (ed & 2u) stamps_cur[t * 4 + 1] = pass + 1;
      if (ed & 4u) stamps_cur[t * 4 + 2] = pass + 1;
      if (ed & 8u) stamps_cur[t * 4 + 3] = pass + 1;
      // plain, idempotent stores into striped words (no same-address atomics on the tile path)
      const uint32_t stripe = (blockIdx.x % NSTRIPE) * STRIPE_STRIDE;
      if (ed & 15u) pf.edge_changed[(pass % COUNTER_RING) * FLAG_SLOT + stripe] = 1u;
      pf.any_change[stripe] = 1u;
    }
  }
}

hipError_t resolve_pass(hipStream_t s, const uint32_t *keys, uint32_t *labels, int h, int w,
                        uint32_t pass, uint32_t *stamps, PassFlags pf) {
  const int tx = tiles_of(w), ty = tiles_of(h);
  k_resolve<<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ty, pass, stamps, pf);
  return hipGetLastError();
}

// Two short runs of flag words into the host's pinned mirror (device-visible host memory) by ONE small launch: inside a
// replayed graph two device-to-host copy nodes were blit kernels of their own with barriers around them, idle time at the
// end of every transform.
__global__ void k_words_to_host(const uint32_t *__restrict__ a, uint32_t na, uint32_t *ha, const uint32_t *__restrict__ b, uint32_t nb,
                                uint32_t *hb) {
  for (uint32_t i = threadIdx.x; i < na; i += blockDim.x) ha[i] = a[i];
  for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) hb[i] = b[i];
}

hipError_t words_to_host(hipStream_t s, const uint32_t *a, uint32_t na, uint32_t *host_a, const uint32_t *b, uint32_t nb, uint32_t *host_b) {
  k_words_to_host<<<1, 64, 0, s>>>(a, na, host_a, b, nb, host_b);
  return hipGetLastError();
}

// ------------------------------------ fused engine: label resolve, two-launch form ----
//
// Same forest as k_resolve, but resolved without iterating over launches:
//   k_resolve_local  every tile builds the parent pointers of its pixels in LDS and collapses the
//                    in-tile part of the forest by pointer jumping (log2(depth) rounds; a pixel whose
//                    parent is a seed -- about half of them on a random field -- takes no part).  A pixel
//                    whose root is a seed of the tile gets that seed's colour; a pixel whose chain
//                    leaves the tile gets a REFERENCE to the halo pixel where it leaves
//                    (REF_BIT | global pixel index) instead of a colour.
//   k_resolve_chase  follows references until a colour is found.  A reference always points to a
//                    pixel with a strictly smaller stamp, so chains end at a seed; concurrent
//                    updates only ever replace a reference by what it resolves to.
// Needs pixel indices < 2^31 (planes up to 46340^2); larger planes use the k_resolve loop.

constexpr uint32_t REF_BIT = 0x80000000u;
constexpr size_t REF_REGION = 64 * 16;          // work-list entries per wave of k_resolve_local: its pixels
constexpr uint32_t REF_ALL = 0xFFFFFFFFu;       // count word of a wave with REF_DENSE references or more: no list (see k_resolve_local)
#ifndef WS_REF_DENSE
#define WS_REF_DENSE 256
#endif
constexpr uint32_t REF_DENSE = WS_REF_DENSE;

typedef uint32_t u32x4_r __attribute__((ext_vector_type(4)));

// Layout: thread t owns the 4 x 4 patch (t & 15, t >> 4) of the 64 x 64 tile, stamps / colours /
// pointers in registers; global accesses are 16 B per lane and row.  The LDS tile has pitch RL_P = 72
// and the patch grid starts at column 4, so that patch rows are 16-byte aligned (ds_*_b128); the halo
// ring sits at rows 0 / 65 and columns 3 / 68.
constexpr int RL_P = 72, RL_X0 = 4, RL_ROWS = TS + 2;
constexpr uint32_t RL_FINAL = 0x8000u;          // pointer flag: the target is a root of the in-tile forest
constexpr uint32_t RL_HALO = 0x4000u;           // pointer flag: ... namely a halo cell, where the chain leaves the tile
static_assert(RL_ROWS * RL_P <= (int)RL_HALO, "cell indices must stay below the flag bits");

// TABLES: the seeds were never painted; their colours come from the side tables of a strictly
// increasing seed list (k_seed_tables) and every pixel of the plane is written here.
// MERGE: the merging transform's final-labels path wants to know, per 64 x 64 tile, whether the tile is one lake
// (every image-interior pixel coloured) and a colour to stand for it -- both are lying around here: tile_min[tile] =
// a colour of the tile that is not a reference (0: not one lake; RL_TILE_UNDECIDED: one lake, but every pixel's
// chain leaves the tile, k_tile_scan looks at the finished labels).  Saves a 268 MB pass over the label plane.
constexpr uint32_t RL_TILE_UNDECIDED = 0xFFFFFFFFu;
// BLOCK: the plane is a row block of a larger field (ws_block_*): its first / last row (halo_flags bit 0 / 1) is a halo
// copy of a neighbour rank's row.  A halo pixel that some flood reached (finite, non-zero stamp) has a colour only its
// owner knows: here it stands for itself as a REFERENCE, so every chain that ends on it is left as a reference to it
// (k_resolve_chase stops at halo pixels) until ws_block_import_boundary has written the halo rows' true colours.
template <bool TABLES, bool MERGE, bool BLOCK>
__global__ __launch_bounds__(NTHREADS, 5) void k_resolve_local(const uint32_t *__restrict__ keys, uint32_t *labels,
                                                            int H, int W, int tilesX, uint32_t *ref_count,
                                                            uint32_t *ref_list, uint32_t max_rounds,
                                                            const uint32_t *__restrict__ seed_mask,
                                                            const uint32_t *__restrict__ word_base, uint32_t *tile_min,
                                                            const uint32_t *__restrict__ gate, int SH, uint32_t *carry_flag,
                                                            const uint32_t *__restrict__ seed_err, int halo_flags) {
  // One LDS tile, used three times: stamps (+ halo ring) -> parent pointers -> painted colours.
  __shared__ __attribute__((aligned(16))) uint32_t sB[RL_ROWS * RL_P];
  // Side tables built from a seed list that turned out not to be strictly increasing, or to leave the plane, describe
  // some other list: nothing may be computed from them (the host repeats the transform once it has read the same words).
  // k_seed_tables has finished before this kernel starts, so the words are final: [0] out of bounds, [2] not strict.
  if (TABLES && seed_err && (seed_err[0] | seed_err[2]) != 0u) return;
  // Speculative launch (ws_segment.hip): queued behind a relaxation pass before the host knows whether that pass still
  // changed anything.  `gate` is the pass's striped convergence slot: any word set -> not a fixpoint yet, leave.
  if (gate && __builtin_amdgcn_ballot_w64(gate[(threadIdx.x & 63) * STRIPE_STRIDE] != 0u) != 0ull) return;
  const uint32_t tile = xcd_span_index(blockIdx.x, gridDim.x);
  const int tile_x = tile % tilesX, tile_y = tile / tilesX;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int pc = tid & 15, pr = tid >> 4;
  const int x0 = tile_x * TS, y0 = tile_y * TS;
  const int gx0 = x0 + pc * 4, gy0 = y0 + pr * 4;
  const int lx0 = RL_X0 + pc * 4, ly0 = 1 + pr * 4;
  WS_STAMP(0);

  // ---- loads: unconditional on clamped addresses (see ws_relax.hip)
  uint32_t K[4][4], Lb[4][4];
  uint32_t seedbits = 0;                         // TABLES: which pixels of the patch are seeds
  uint32_t tab_mask[4] = {0u, 0u, 0u, 0u}, tab_base[4] = {0u, 0u, 0u, 0u};   // TABLES, vec: the table words of the 4 patch rows
  const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(labels)) & 15u) == 0;
  if (vec) {          // W % 4 == 0: a patch is wholly inside or wholly outside the plane in x
    const int gxc = min(gx0, W - 4);
    u32x4_r kv[4], lv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t g = (size_t)min(gy0 + r, H - 1) * W + gxc;
      kv[r] = *reinterpret_cast<const u32x4_r *>(keys + g);
      if (TABLES) {      // requested with the stamps (a late load would stall the whole workgroup after the rounds);
        tab_mask[r] = seed_mask[g >> 5];      // 8 registers across the jumping rounds instead of 16 colours
        tab_base[r] = word_base[g >> 5];
        lv[r] = u32x4_r{0u, 0u, 0u, 0u};
      } else {
        lv[r] = *reinterpret_cast<const u32x4_r *>(labels + g);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      K[r][0] = kv[r].x; K[r][1] = kv[r].y; K[r][2] = kv[r].z; K[r][3] = kv[r].w;
      Lb[r][0] = lv[r].x; Lb[r][1] = lv[r].y; Lb[r][2] = lv[r].z; Lb[r][3] = lv[r].w;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const size_t g = (size_t)min(gy0 + r, H - 1) * W + min(gx0 + c, W - 1);
        K[r][c] = keys[g];
        Lb[r][c] = TABLES ? 0u : labels[g];
      }
  }
  {
    // halo ring: thread t loads (top, bottom, left, right)[t & 63]
    const int side = tid >> 6, t = tid & 63;
    const int hy = side == 0 ? y0 - 1 : (side == 1 ? y0 + TS : y0 + t);
    const int hx = side == 2 ? x0 - 1 : (side == 3 ? x0 + TS : x0 + t);
    const size_t hg = (size_t)min(max(hy, 0), H - 1) * W + min(max(hx, 0), W - 1);
    const uint32_t hv = keys[hg];
    const bool hok = hy >= 0 && hy < H && hx >= 0 && hx < W;
    // (workgroup uniform: a tile that lies wholly inside the plane has nothing to mask -- this kernel is bound by vector issue)
    if (!(x0 + TS <= W && y0 + TS <= H)) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (!(gy0 + r < H && gx0 + c < W)) { K[r][c] = KEY_INF; Lb[r][c] = 0u; }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (TABLES && K[r][c] == 0u) seedbits |= 1u << (r * 4 + c);      // only a seed (stamp 0) has a colour of its own
      *reinterpret_cast<u32x4_r *>(&sB[(ly0 + r) * RL_P + lx0]) = u32x4_r{K[r][0], K[r][1], K[r][2], K[r][3]};
    }
    sB[(hy - (y0 - 1)) * RL_P + (hx - (x0 - 1)) + (RL_X0 - 1)] = hok ? hv : KEY_INF;
  }
  __syncthreads();
  WS_STAMP(1);

  // ---- parent pointer of every pixel of the patch.  A pointer carries RL_FINAL when its target is a
  // root of the in-tile forest (a seed, or a halo cell = where the chain leaves the tile): such a
  // pixel never enters the jumping rounds.  Roots point at themselves.
  uint32_t P[4][4];
  uint32_t carried = 0xFFFFFFFFu;
  uint32_t live = 0;                             // pixels whose pointer may still move
  {
    const u32x4_r up4 = *reinterpret_cast<const u32x4_r *>(&sB[(ly0 - 1) * RL_P + lx0]);
    const u32x4_r dn4 = *reinterpret_cast<const u32x4_r *>(&sB[(ly0 + 4) * RL_P + lx0]);
    const uint32_t up[4] = {up4.x, up4.y, up4.z, up4.w}, dn[4] = {dn4.x, dn4.y, dn4.z, dn4.w};
    uint32_t Lc[4], Rc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { Lc[r] = sB[(ly0 + r) * RL_P + lx0 - 1]; Rc[r] = sB[(ly0 + r) * RL_P + lx0 + 4]; }
    // Branch-free on purpose, 0/1 words and selects only: written with `if` / `&&` this block compiled
    // to ~75 instructions per pixel of exec-mask juggling and was the longest phase of the kernel.
    const uint32_t top_halo = pr == 0, bot_halo = pr == 15, left_halo = pc == 0, right_halo = pc == 15;
    uint32_t row_int[4], col_int[4];
    const int ry0 = SH == H ? gy0 : gy0 % SH;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ry = ry0 + i >= SH ? ry0 + i - SH : ry0 + i;       // row inside its slice (SH == H: one image)
      row_int[i] = (uint32_t)(ry >= 1) & (uint32_t)(ry < SH - 1) & (uint32_t)(gy0 + i < H);
      col_int[i] = (uint32_t)(gx0 + i >= 1) & (uint32_t)(gx0 + i < W - 1);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const uint32_t cell = (uint32_t)((ly0 + r) * RL_P + lx0 + c);
        const uint32_t k = K[r][c];
        // flooded pixels are interior pixels (lib.rs:220-222) with a finite, non-seed stamp.  The relaxation never lowers the
        // stamp of a border pixel of the image (or of a slice of a stack: its base is KEY_INF), so such a pixel is a seed or
        // never coloured and the stamp alone says "not flooded"; only a row block's halo rows hold finite stamps on a border
        // row (a neighbour rank's): BLOCK keeps the masks.
        const uint32_t flood = BLOCK ? (uint32_t)(k - 1u < KEY_INF - 1u) & row_int[r] & col_int[c] : (uint32_t)(k - 1u < KEY_INF - 1u);
        // a finite stamp of a flooded pixel with ring 0: a carry out of the 24-bit ring field (the relaxation of a whole
        // transform leaves this test to the one kernel that reads the finished plane)
        // (smallest, over the flooded pixels, of the ring field moved to the top of the word: 0 = a carry; two ops per pixel)
        carried = min(carried, (k << 8) | (flood ^ 1u));
        const uint32_t d = r == 3 ? dn[c] : K[r + 1][c];
        const uint32_t rr = c == 3 ? Rc[r] : K[r][c + 1];
        const uint32_t l = c == 0 ? Lc[r] : K[r][c - 1];
        const uint32_t u = r == 0 ? up[c] : K[r - 1][c];
        // first earlier neighbour in D,R,L,U (lib.rs:190, 245); "up" is the fall-through: at a
        // fixpoint one of the four is earlier
        const uint32_t td = d < k, tr = rr < k, tl = l < k;
        const uint32_t tgt = cell + (uint32_t)(td ? RL_P : (tr ? 1 : (tl ? -1 : -RL_P)));
        const uint32_t kp = td ? d : (tr ? rr : (tl ? l : u));
        // does the chosen step leave the tile?  (r, c are compile-time: most terms vanish)
        uint32_t out_of_tile = 0;
        if (r == 3) out_of_tile |= td & bot_halo;
        if (c == 3) out_of_tile |= (td ^ 1u) & tr & right_halo;
        if (c == 0) out_of_tile |= (td ^ 1u) & (tr ^ 1u) & tl & left_halo;
        if (r == 0) out_of_tile |= (td ^ 1u) & (tr ^ 1u) & (tl ^ 1u) & top_halo;
        const uint32_t fin = (uint32_t)(kp == 0u) | out_of_tile;
        const uint32_t moving = tgt | (fin ? RL_FINAL : 0u) | (out_of_tile ? RL_HALO : 0u);
        P[r][c] = flood ? moving : (cell | RL_FINAL);
        live |= (flood & (fin ^ 1u)) << (r * 4 + c);
      }
    }
  }
  if (carry_flag && carried == 0u) atomicExch(carry_flag, 1u);      // never taken on sane inputs
  __syncthreads();                               // every stamp has been read: the tile becomes pointers
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<u32x4_r *>(&sB[(ly0 + r) * RL_P + lx0]) = u32x4_r{P[r][0], P[r][1], P[r][2], P[r][3]};
  __syncthreads();
  WS_STAMP(2);

  // Pointer jumping, P <- P(P), until P's target is a root -- WITHOUT rounds: every value a pixel can
  // read from the tile, old or freshly compressed by its owner, is a pointer to one of its ancestors
  // (flagged when that ancestor is a root), so threads neither wait for each other nor agree on
  // when to stop; a thread is done when its own 16 pointers are final.  One barrier after the loop
  // (the tile is reused for colours), none inside.
  for (uint32_t it = 0; live != 0 && it < max_rounds; ++it) {       // max_rounds: timing experiments only (WS_DEBUG_MAXIT)
    // (tried: all 16 reads issued up front on clamped cells, no per-pixel branch -- the extra LDS traffic
    // of the pixels that are done cost more than the serialised waits)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if ((live >> (r * 4 + c)) & 1u) {
          const uint32_t g = sB[P[r][c]];        // live pointers carry no flag: the value is the cell index
          P[r][c] = g;
          sB[(ly0 + r) * RL_P + lx0 + c] = g;
          if (g & RL_FINAL) live &= ~(1u << (r * 4 + c));
        }
      }
    WS_STAMP_VALUE(6, it + 1);
  }
  __syncthreads();
  WS_STAMP(3);
  if (TABLES) {
    // seed colours from the side tables; with W % 4 == 0 a patch row sits in one mask word
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gyc = min(gy0 + r, H - 1);
      if (vec) {
        const size_t g = (size_t)gyc * W + min(gx0, W - 4);
        // (the asm pins the colour arithmetic here: computed early it holds 16 registers through the rounds)
        asm volatile("" : "+v"(tab_mask[r]), "+v"(tab_base[r]));
        const uint32_t mw = tab_mask[r], wb = tab_base[r], sh = (uint32_t)(g & 31u);
        // colour of the row's first seed; the seeds after it in the row count on from there (one table popcount per
        // row instead of one per pixel: the kernel is VALU-bound)
        const uint32_t first = wb + __popc(mw & ((1u << sh) - 1u)) + 1u;
        const uint32_t nib = (seedbits >> (r * 4)) & 15u;
        Lb[r][0] = nib & 1u ? first : 0u;
        Lb[r][1] = nib & 2u ? first + (nib & 1u) : 0u;
        Lb[r][2] = nib & 4u ? first + __popc(nib & 3u) : 0u;
        Lb[r][3] = nib & 8u ? first + __popc(nib & 7u) : 0u;
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const size_t g = (size_t)gyc * W + min(gx0 + c, W - 1);
          const uint32_t col = word_base[g >> 5] + __popc(seed_mask[g >> 5] & ((1u << (g & 31u)) - 1u)) + 1u;
          Lb[r][c] = (seedbits >> (r * 4 + c)) & 1u ? col : 0u;
        }
      }
    }
  }
  if (BLOCK && TABLES) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int gy = gy0 + r;
      const bool halo_row = (gy == 0 && (halo_flags & 1)) || (gy == H - 1 && (halo_flags & 2));
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (halo_row && gx0 + c < W && K[r][c] != 0u && K[r][c] != KEY_INF) Lb[r][c] = REF_BIT | (uint32_t)((size_t)gy * W + gx0 + c);
    }
  }
  // every pointer is final and in registers: the tile now becomes COLOURS -- a pixel's own colour (a seed's, or
  // none), and in the halo ring a REFERENCE to the global pixel the cell stands for -- so that whatever a
  // pointer ends at, the answer is one unconditional LDS read of its target: no decode of the cell index, no
  // select chain, no branch (this phase was 39 instructions per pixel with them)
  {
    const int side = tid >> 6, t = tid & 63;
    const int hy = side == 0 ? y0 - 1 : (side == 1 ? y0 + TS : y0 + t);
    const int hx = side == 2 ? x0 - 1 : (side == 3 ? x0 + TS : x0 + t);
    const int hyc = min(max(hy, 0), H - 1), hxc = min(max(hx, 0), W - 1);
    // (tried: a halo cell that is a SEED holds the seed's colour, from the side tables, instead of a reference -- half of the
    // leaving chains of a random field end there: k_resolve_chase 41 -> 27 us, and this kernel 170 -> 183 us for the two
    // table words per halo cell, the left and right columns' a cache line each)
    sB[(hy - (y0 - 1)) * RL_P + (hx - (x0 - 1)) + (RL_X0 - 1)] = REF_BIT | (uint32_t)((size_t)hyc * W + hxc);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r)
    *reinterpret_cast<u32x4_r *>(&sB[(ly0 + r) * RL_P + lx0]) = u32x4_r{Lb[r][0], Lb[r][1], Lb[r][2], Lb[r][3]};
  __syncthreads();

  uint32_t refmask = 0;                          // pixels whose chain leaves the tile
  uint32_t out[4][4];
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) out[r][c] = sB[P[r][c] & (RL_HALO - 1u)];      // 16 independent reads, one wait
  const bool whole = x0 + TS <= W && y0 + TS <= H;      // workgroup uniform
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gy = gy0 + r;
    if (whole) {
#pragma unroll
      for (int c = 0; c < 4; ++c) refmask |= (out[r][c] >> 31) << (r * 4 + c);      // REF_BIT is the top bit
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        refmask |= ((uint32_t)((out[r][c] & REF_BIT) != 0u) & (uint32_t)(gy < H) & (uint32_t)(gx0 + c < W)) << (r * 4 + c);
    }
    if (gy < H) {
      if (vec) {
        if (gx0 < W) *reinterpret_cast<u32x4_r *>(labels + (size_t)gy * W + gx0) = u32x4_r{out[r][0], out[r][1], out[r][2], out[r][3]};
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (gx0 + c < W && (TABLES || (P[r][c] & (RL_HALO - 1u)) != (uint32_t)((ly0 + r) * RL_P + lx0 + c)))
            labels[(size_t)gy * W + gx0 + c] = out[r][c];
      }
    }
  }
  if (MERGE) {
    // min and max of (label - 1) over the patch: an uncoloured pixel (0) wraps to the top, a reference lies at or
    // above REF_BIT - 1; so max == ~0 says "a hole", min < REF_BIT - 1 is a colour that needs no chase
    __shared__ uint32_t sTileLo[NTHREADS / 64], sTileHi[NTHREADS / 64], sTileAny[NTHREADS / 64];
    const bool border_tile = x0 == 0 || y0 == 0 || x0 + TS >= W || y0 + TS >= H;      // uniform
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    bool any_interior = !border_tile;
    if (!border_tile) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) { const uint32_t t = out[r][c] - 1u; lo = min(lo, t); hi = max(hi, t); }
    } else {
      // image-border pixels never flood (the flood step only visits 3x3 window centres, lib.rs:220-222; edge correction pads
      // with zeros, lib.rs:1340-1352, whose ring is then the border): only interior pixels decide "one lake";
      // a corner pixel touches no interior pixel, its colour never merges and must not stand for the tile
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int gy = gy0 + r, gx = gx0 + c;
          const bool in_plane = gy < H && gx < W;
          const bool inter = gy >= 1 && gy < H - 1 && gx >= 1 && gx < W - 1;
          const bool corner = (gy == 0 || gy == H - 1) && (gx == 0 || gx == W - 1);
          const uint32_t t = out[r][c] - 1u;
          if (in_plane && !corner) lo = min(lo, t);
          if (inter) { hi = max(hi, t); any_interior = true; }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
      lo = min(lo, (uint32_t)__shfl_xor(lo, o, 64));
      hi = max(hi, (uint32_t)__shfl_xor(hi, o, 64));
    }
    const bool wave_any = __ballot(any_interior) != 0ull;
    if (lane == 0) { sTileLo[wave] = lo; sTileHi[wave] = hi; sTileAny[wave] = wave_any ? 1u : 0u; }
    __syncthreads();
    if (tid == 0) {
      uint32_t l = sTileLo[0], h2 = sTileHi[0], a = sTileAny[0];
#pragma unroll
      for (int k = 1; k < NTHREADS / 64; ++k) { l = min(l, sTileLo[k]); h2 = max(h2, sTileHi[k]); a |= sTileAny[k]; }
      const bool one_lake = h2 != 0xFFFFFFFFu && a != 0u;
      tile_min[tile] = !one_lake ? 0u : (l < REF_BIT - 1u ? l + 1u : RL_TILE_UNDECIDED);
    }
  }
  WS_STAMP(4);
  // work list of the reference pixels for k_resolve_chase: every WAVE owns a fixed region of the list
  // (64 lanes x 16 pixels) and a count word -- no reservation atomic, no barrier, no cross-wave offsets
  // (a returning atomicAdd per workgroup kept all four waves waiting ~1.5 us)
  const uint32_t cnt = __popc(refmask);
  const uint32_t incl = wave_inclusive_sum(cnt);
  const size_t region = (size_t)blockIdx.x * (NTHREADS / 64) + wave;
  // A quarter or more of the wave's pixels references (a smooth map: all of them, in a tile without a seed): no list --
  // k_resolve_chase reads the wave's 64 x 16 pixels from the label plane itself, 16 bytes a lane and row, instead of 8
  // bytes of list per reference written here and read there, and stores whole rows instead of single words (at 8192^2 and
  // a correlation length of 64 px the lists were 1 GB of a transform's traffic: resolve 1.05 -> 0.44 ms).
  const bool all_refs = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63) >= REF_DENSE;      // wave uniform
  if (lane == 63) ref_count[region] = all_refs ? REF_ALL : incl;
  if (refmask && !all_refs) {      // (the list writes are 8 of this kernel's 165 us on the bench field)
    // (pixel, what it refers to): the chase starts at the target without reading the pixel's own label first
    uint2 *dst = reinterpret_cast<uint2 *>(ref_list) + region * REF_REGION + (incl - cnt);
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((refmask >> i) & 1u) *dst++ = make_uint2((uint32_t)((size_t)(gy0 + (i >> 2)) * W + gx0 + (i & 3)), out[i >> 2][i & 3] & ~REF_BIT);
  }
  WS_STAMP(5);
}

// A wave per CH_R consecutive list regions (grid-stride).  A reference always points to a pixel with a strictly
// smaller stamp, so chains end at a seed; a racing reader sees either the reference or what it
// resolves to.
// On the bench field a region holds ~20 references and nearly every chain is one hop long: the kernel is three dependent
// memory round trips (count, entry, target) per wave and nothing else, so a wave takes the counts of CH_R regions in one
// load, walks their entries as ONE flattened list, CH_U entries per lane in flight, and issues the first hop of all of
// them before it waits (one region at a time, four per wave one after the other: 48 us at 8192^2, 75 MB of traffic; this
// form 41 us, whatever CH_R and CH_U: reading the lists is 6 us of it, the scattered 4-byte loads of the first hop ~10-17 and
// the scattered 4-byte stores ~18-28 -- partial writes into lines k_resolve_local has just sent to HBM; profiles/r2_v7_chase_ab.log).
// (groups of 8 regions balance badly on smooth maps, where every pixel is a reference and a region holds 1024: 8192^2 at
// correlation 64 / 256 px 344 / 212 us against 301 / 144 with 4 and 269 / 259 for the old one-region walk; bench field 38 us either way)
constexpr int CH_R = 4, CH_U = 4;

// Follows the chain that starts with the reference `v` (the label some pixel holds) to its colour -- or to the first
// reference that leads out of [follow_from, follow_from + span).  COMPRESS (planes whose chains cross many tiles: smooth
// maps): a chain of more than two hops that ended in a colour is walked a second time and every pixel on it is given that
// colour -- its own final label, so a plain store that no owner's store can contradict -- and whoever comes by later stops
// there.  (Path halving by compare-and-swap was measured first: the chains are shared by thousands of waves, the swaps hit
// the same addresses, and same-address atomics retire at ~25 M/s: the chase took twice as long, 0.43 -> 0.77 ms.)
template <bool COMPRESS>
__device__ __forceinline__ uint32_t follow_chain(uint32_t *labels, uint32_t v, size_t follow_from, size_t span, size_t n) {
  const uint32_t start = v;
  size_t hops = 1;
  for (; (v & REF_BIT) && (size_t)(v & ~REF_BIT) - follow_from < span && hops < n; ++hops)
    v = __hip_atomic_load(labels + (v & ~REF_BIT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (COMPRESS && hops > 3 && !(v & REF_BIT)) {
    uint32_t w = start;
    for (size_t k = 1; k < hops && (w & REF_BIT) && (size_t)(w & ~REF_BIT) - follow_from < span; ++k) {
      const uint32_t q = w & ~REF_BIT;
      w = __hip_atomic_load(labels + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (w & REF_BIT) __hip_atomic_store(labels + q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return v;
}

__global__ __launch_bounds__(256) void k_resolve_chase(uint32_t *labels, const uint32_t *__restrict__ ref_count,
                                const uint32_t *__restrict__ ref_list, size_t nregions, size_t n,
                                const uint32_t *__restrict__ gate, const uint32_t *__restrict__ seed_err,
                                size_t follow_from, size_t follow_to, int H, int W, int tilesX, uint32_t ntiles, int halve) {
  // [follow_from, follow_to): the pixels a chain may be followed THROUGH -- the whole plane [0, n), or, for a row block
  // whose halo rows still hold references to themselves, the plane without those rows: a chain stops at a halo pixel.
  const int lane = threadIdx.x & 63;
  if (seed_err && (seed_err[0] | seed_err[2]) != 0u) return;      // invalid side tables: k_resolve_local wrote no lists (see there)
  if (gate && __builtin_amdgcn_ballot_w64(gate[lane * STRIPE_STRIDE] != 0u) != 0ull) return;      // see k_resolve_local
  const size_t wave0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  const size_t ngroups = (nregions + CH_R - 1) / CH_R;
  const size_t span = follow_to - follow_from;
  for (size_t grp = wave0; grp < ngroups; grp += nwaves) {
    // (last region first: the labels k_resolve_local wrote last are the ones still in the memory-side cache -- 43 -> 38 us)
    const size_t r0 = (ngroups - 1 - grp) * CH_R;
    uint32_t cnt = 0;
    if (lane < CH_R && r0 + lane < nregions) cnt = ref_count[r0 + lane];
    const unsigned long long all_refs = __builtin_amdgcn_ballot_w64(cnt == REF_ALL);      // regions without a list (k_resolve_local)
    cnt = cnt == REF_ALL ? 0u : min(cnt, (uint32_t)REF_REGION);
    const uint32_t incl = wave_inclusive_sum(cnt);      // lanes >= CH_R: the total
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    uint32_t ends[CH_R - 1];                             // uniform: where region i + 1 starts in the flattened list
#pragma unroll
    for (int i = 0; i < CH_R - 1; ++i) ends[i] = (uint32_t)__builtin_amdgcn_readlane((int)incl, i);
    for (uint32_t jb = 0; jb < tot; jb += 64 * CH_U) {
      uint2 e[CH_U];
      uint32_t v[CH_U];
      bool ok[CH_U];
#pragma unroll
      for (int u = 0; u < CH_U; ++u) {
        const uint32_t j = jb + u * 64 + lane;
        ok[u] = j < tot;
        uint32_t reg = 0, base = 0;
#pragma unroll
        for (int i = 0; i < CH_R - 1; ++i)
          if (j >= ends[i]) { reg = i + 1; base = ends[i]; }
        // (a lane past the end reads entry 0 of the group's first region: inside the list, ignored)
        e[u] = reinterpret_cast<const uint2 *>(ref_list)[(r0 + (ok[u] ? reg : 0u)) * REF_REGION + (ok[u] ? j - base : 0u)];
      }
      // A reference points at a pixel with a strictly smaller stamp, so a chain visits a pixel at most once: fewer than n
      // hops.  The range test and the hop bound are belt and braces: with invalid side tables (the only source of words
      // that are not colours or references of this plane) both resolve kernels have already left.
      // first hop of every entry: unconditional loads (pixel 0 for a lane with nothing to follow), one wait
      bool follow[CH_U];
      uint32_t first[CH_U];
#pragma unroll
      for (int u = 0; u < CH_U; ++u) {
        v[u] = e[u].y | REF_BIT;
        follow[u] = ok[u] && n != 0 && (size_t)e[u].y - follow_from < span;
        first[u] = __hip_atomic_load(labels + (follow[u] ? e[u].y : (uint32_t)follow_from), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int u = 0; u < CH_U; ++u) {
        if (follow[u]) v[u] = follow_chain<false>(labels, first[u], follow_from, span, n);      // (lists: short chains, a hop or two)
      }
#pragma unroll
      for (int u = 0; u < CH_U; ++u)
        if (ok[u]) labels[e[u].x] = v[u];
    }
    // the regions without a list: the wave of k_resolve_local that owned region r held the 4 x 4 patches (lane & 15,
    // 4 (r & 3) + lane / 16) of tile xcd_span_index(r / 4); a label that is no reference is left alone
    for (unsigned long long m = all_refs; m != 0ull; m &= m - 1ull) {
      const size_t region = r0 + (size_t)__builtin_ctzll(m);
      const uint32_t tile = xcd_span_index((uint32_t)(region / (NTHREADS / 64)), ntiles);
      const int gx0 = (int)(tile % (uint32_t)tilesX) * TS + (lane & 15) * 4;
      const int gy0 = (int)(tile / (uint32_t)tilesX) * TS + ((int)(region % (NTHREADS / 64)) * 4 + (lane >> 4)) * 4;
      const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(labels) & 15u) == 0;
      // (pixels outside the plane -- a tile at its right or lower edge -- read as label 0: no reference, never stored)
      uint32_t L[4][4];
      if (vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          u32x4_r q = u32x4_r{0u, 0u, 0u, 0u};
          if (gy0 + r < H && gx0 < W) q = *reinterpret_cast<const u32x4_r *>(labels + (size_t)(gy0 + r) * W + gx0);
          L[r][0] = q.x; L[r][1] = q.y; L[r][2] = q.z; L[r][3] = q.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) L[r][c] = gy0 + r < H && gx0 + c < W ? labels[(size_t)(gy0 + r) * W + gx0 + c] : 0u;
      }
      // first hops: sixteen unconditional loads, one wait; neighbouring pixels mostly leave the tile by the same halo pixel
      uint32_t first[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t t = L[r][c] & ~REF_BIT;
          const bool go = (L[r][c] & REF_BIT) && n != 0 && (size_t)t - follow_from < span;
          first[r][c] = __hip_atomic_load(labels + (go ? t : (uint32_t)follow_from), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      // ... and the rest of a chain once per run of pixels that share it (the hop bound as above)
      uint32_t memo_from = 0u, memo_to = 0u;      // (0 is not a reference: never matches)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const uint32_t t = L[r][c] & ~REF_BIT;
          if ((L[r][c] & REF_BIT) && n != 0 && (size_t)t - follow_from < span) {
            uint32_t v = first[r][c];
            if ((v & REF_BIT) && (size_t)(v & ~REF_BIT) - follow_from < span) {
              if (v == memo_from) v = memo_to;
              else {
                const uint32_t from = v;
                v = halve ? follow_chain<true>(labels, v, follow_from, span, n) : follow_chain<false>(labels, v, follow_from, span, n);
                memo_from = from; memo_to = v;
              }
            }
            L[r][c] = v;
          }
        }
      if (vec) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (gy0 + r < H && gx0 < W) *reinterpret_cast<u32x4_r *>(labels + (size_t)(gy0 + r) * W + gx0) = u32x4_r{L[r][0], L[r][1], L[r][2], L[r][3]};
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (gy0 + r < H && gx0 + c < W) labels[(size_t)(gy0 + r) * W + gx0 + c] = L[r][c];
      }
    }
  }
}

// words of scratch the two-launch resolve needs: a count and a REF_REGION-entry list per wave of k_resolve_local
size_t resolve_ref_capacity(int h, int w) {
  const size_t nregions = (size_t)tiles_of(w) * tiles_of(h) * (NTHREADS / 64);
  return nregions * (1 + 2 * REF_REGION);      // a count and REF_REGION (pixel, target) pairs per wave
}

hipError_t resolve_two_launch(hipStream_t s, const uint32_t *keys, uint32_t *labels, int h, int w, uint32_t *ref_scratch,
                              uint32_t max_rounds, const uint32_t *seed_mask, const uint32_t *word_base, uint32_t *tile_min,
                              const uint32_t *gate, int slice_h, uint32_t *carry_flag, const uint32_t *seed_err, int halo_flags) {
  const int tx = tiles_of(w), ty = tiles_of(h);
  const int sh = slice_h > 0 ? slice_h : h;
  const size_t n = (size_t)h * w;
  if (n == 0) return hipSuccess;
  const size_t nregions = (size_t)tx * ty * (NTHREADS / 64);
  uint32_t *ref_count = ref_scratch, *ref_list = ref_scratch + nregions;
  // halo_flags 4 (a tile of a field cut in both directions, painted labels): the plane's border RING holds other ranks' pixels
  // with whatever label is known of them so far -- not flooded here (the BLOCK masks), roots like seeds; chains end there
  const bool ring = !seed_mask && (halo_flags & 4) != 0;
  halo_flags &= 3;
  if (ring)
    k_resolve_local<false, false, true><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, nullptr, nullptr, nullptr, gate, sh, carry_flag, nullptr, 0);
  else if (seed_mask && halo_flags)
    k_resolve_local<true, false, true><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, seed_mask, word_base, nullptr, gate, sh, carry_flag, seed_err, halo_flags);
  else if (seed_mask && tile_min)
    k_resolve_local<true, true, false><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, seed_mask, word_base, tile_min, gate, sh, carry_flag, seed_err, 0);
  else if (seed_mask)
    k_resolve_local<true, false, false><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, seed_mask, word_base, nullptr, gate, sh, carry_flag, seed_err, 0);
  else if (tile_min)
    k_resolve_local<false, true, false><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, nullptr, nullptr, tile_min, gate, sh, carry_flag, nullptr, 0);
  else
    k_resolve_local<false, false, false><<<tx * ty, NTHREADS, 0, s>>>(keys, labels, h, w, tx, ref_count, ref_list, max_rounds, nullptr, nullptr, nullptr, gate, sh, carry_flag, nullptr, 0);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const unsigned grid = (unsigned)std::min<size_t>(((nregions + CH_R - 1) / CH_R + 3) / 4, 4096);
  const size_t from = (halo_flags & 1) ? (size_t)w : 0, to = (halo_flags & 2) ? n - (size_t)w : n;
  k_resolve_chase<<<grid, 256, 0, s>>>(labels, ref_count, ref_list, nregions, n, gate, seed_mask ? seed_err : nullptr, from, to, h, w, tx, (uint32_t)(tx * ty), halo_flags == 0 && !ring ? 1 : 0);
  return hipGetLastError();
}

// the chase alone, over the lists the last resolve_two_launch left in ref_scratch: every chain is followed to its end
// (a row block after ws_block_import_boundary has put the neighbours' colours into its halo rows)
hipError_t resolve_chase_again(hipStream_t s, uint32_t *labels, int h, int w, uint32_t *ref_scratch) {
  const size_t n = (size_t)h * w;
  if (n == 0) return hipSuccess;
  const size_t nregions = (size_t)tiles_of(w) * tiles_of(h) * (NTHREADS / 64);
  const unsigned grid = (unsigned)std::min<size_t>(((nregions + CH_R - 1) / CH_R + 3) / 4, 4096);
  k_resolve_chase<<<grid, 256, 0, s>>>(labels, ref_scratch, ref_scratch + nregions, nregions, n, nullptr, nullptr, 0, n, h, w, tiles_of(w), (uint32_t)(tiles_of(w) * tiles_of(h)), 0);
  return hipGetLastError();
}

// ------------------------------------------------ sweep engine: one flood step -------
//
// lib.rs:196-257 one-to-one: every interior pixel that is flooded (img <= level),
// uncoloured and has a coloured 4-neighbour takes the first coloured neighbour in
// down,right,left,up order; reads only the previous plane (lin), writes the next (lout).

// first coloured neighbour in down, right, left, up order (lib.rs:190, 237-248 with col0)
__device__ __forceinline__ uint32_t pick_drlu(uint32_t d, uint32_t r, uint32_t l, uint32_t u) {
  return d ? d : (r ? r : (l ? l : u));
}

// scalar form: any width / alignment.  All loads unconditional on clamped addresses; the "something
// was coloured" word is a plain idempotent store (one shared atomic per workgroup would serialise).
__global__ __launch_bounds__(256) void k_flood_step(const uint8_t *__restrict__ img, size_t img_stride,
                                                    const uint32_t *__restrict__ lin, uint32_t *__restrict__ lout,
                                                    int H, int W, uint32_t level, uint32_t *counter, int pad) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int xc = min(x, W - 1), yc = min(y, H - 1);
  const int xm = max(xc - 1, 0), xp = min(xc + 1, W - 1), ym = max(yc - 1, 0), yp = min(yc + 1, H - 1);
  const uint32_t c = lin[(size_t)yc * W + xc];
  const uint32_t d = lin[(size_t)yp * W + xc], r = lin[(size_t)yc * W + xp];
  const uint32_t l = lin[(size_t)yc * W + xm], u = lin[(size_t)ym * W + xc];
  const uint32_t v = img[pad ? padded_img_index(yc, xc, W, H, img_stride) : (size_t)yc * img_stride + xc];      // pad: virtual ring of zeros (ws_common.hpp)
  const bool floodable = c == 0u && y >= 1 && y < H - 1 && x >= 1 && x < W - 1 && v <= level;   // lib.rs:224-226
  const uint32_t pick = floodable ? pick_drlu(d, r, l, u) : 0u;
  if (x < W && y < H) lout[(size_t)y * W + x] = c ? c : pick;
  if (pick && x < W && y < H) *counter = 1u;
}

// vector form (W % 4 == 0, image rows dword aligned): a thread owns 4 consecutive pixels; labels move
// as 16-byte vectors, the image as one dword; a workgroup covers 1024 pixels of one row
__global__ __launch_bounds__(256) void k_flood_step4(const uint8_t *__restrict__ img, size_t img_stride,
                                                     const uint32_t *__restrict__ lin, uint32_t *__restrict__ lout,
                                                     int H, int W, uint32_t level, uint32_t *counter) {
  const int y = blockIdx.y;
  const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x0 >= W) return;
  const int ym = max(y - 1, 0), yp = min(y + 1, H - 1);
  const u32x4_r c = *reinterpret_cast<const u32x4_r *>(lin + (size_t)y * W + x0);
  const u32x4_r dn = *reinterpret_cast<const u32x4_r *>(lin + (size_t)yp * W + x0);
  const u32x4_r up = *reinterpret_cast<const u32x4_r *>(lin + (size_t)ym * W + x0);
  const uint32_t lft = lin[(size_t)y * W + max(x0 - 1, 0)], rgt = lin[(size_t)y * W + min(x0 + 4, W - 1)];
  const uint32_t iv = *reinterpret_cast<const uint32_t *>(img + (size_t)y * img_stride + x0);
  const bool row_int = y >= 1 && y < H - 1;
  const uint32_t cc[4] = {c.x, c.y, c.z, c.w}, dd[4] = {dn.x, dn.y, dn.z, dn.w}, uu[4] = {up.x, up.y, up.z, up.w};
  const uint32_t ll[4] = {lft, c.x, c.y, c.z}, rr[4] = {c.y, c.z, c.w, rgt};
  uint32_t out[4], any = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + k;
    const bool floodable = cc[k] == 0u && row_int && x >= 1 && x < W - 1 && ((iv >> (8 * k)) & 0xFFu) <= level;
    const uint32_t pick = floodable ? pick_drlu(dd[k], rr[k], ll[k], uu[k]) : 0u;
    out[k] = cc[k] ? cc[k] : pick;
    any |= pick;
  }
  *reinterpret_cast<u32x4_r *>(lout + (size_t)y * W + x0) = u32x4_r{out[0], out[1], out[2], out[3]};
  if (any) *counter = 1u;
}

hipError_t flood_step(hipStream_t s, const uint8_t *img, size_t img_stride, const uint32_t *lin,
                      uint32_t *lout, int h, int w, uint32_t level, uint32_t *counter, bool padded) {
  if (h == 0 || w == 0) return hipSuccess;
  if (!padded && (w & 3) == 0 && ((reinterpret_cast<uintptr_t>(img) | img_stride) & 3u) == 0) {
    dim3 grid((w / 4 + 255) / 256, h);
    k_flood_step4<<<grid, 256, 0, s>>>(img, img_stride, lin, lout, h, w, level, counter);
  } else {
    dim3 grid((w + 63) / 64, (h + 3) / 4);
    k_flood_step<<<grid, 256, 0, s>>>(img, img_stride, lin, lout, h, w, level, counter, padded ? 1 : 0);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------- find_local_minima --------
//
// lib.rs:1178-1197: interior pixels whose 8 neighbours are all strictly smaller, emitted in row-major order.
// Two launches: the COUNT pass looks at the image once and leaves the 4-bit answer of every 4-pixel group in a nibble plane
// and every row's place in the list (row counts, scanned by its last workgroup); the WRITE pass turns nibbles into list
// entries -- and, when asked, into the seed side tables of that list (k_seed_tables' layout) -- without reading the image again.
//
// The count pass is vector-issue work (8192^2 random bytes: 73 us when every pixel took the max of its eight neighbours
// byte by byte, ~37 instructions a pixel), so it computes in packed 16-bit pairs and reuses row maxima:
//     h3[y][x] = max(r[y][x-1], r[y][x], r[y][x+1])      m2[y][x] = max(r[y][x-1], r[y][x+1])
//     strict maximum at (y, x)  <=>  r[y][x] > max(h3[y-1][x], h3[y+1][x], m2[y][x])
// a thread owns 4 columns of MIN_RS rows: per row 3 loads, 5 byte permutes, 6 packed maxima for h3 / m2 and 11 more
// operations for the four answers -- ~6 a pixel.  Rows and columns outside the image are CLAMPED to the nearest one inside:
// a pixel on the image border then meets its own value among its "neighbours" and can never be a strict maximum, which is
// the reference's "interior only" (3 x 3 windows, lib.rs:1183) without a single mask.

constexpr int SEG = 1024;            // pixels per workgroup step: 256 threads x 4
constexpr int MIN_RS = 8;            // rows per workgroup of the count pass
static_assert(MIN_RS == 8, "k_minima_count's strip sum and k_minima_write's offsets assume strips of eight rows");

size_t minima_segments(int h, int w) { return (size_t)h + (h + MIN_RS - 1) / MIN_RS + 1; }     // one count per row, one per strip of MIN_RS rows
size_t minima_mask_bytes(int h, int w) { return (size_t)h * ((w + SEG - 1) / SEG) * 256; }    // one nibble byte per 4-pixel group

typedef unsigned short u16x2_m __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2_m, a), __builtin_bit_cast(u16x2_m, b)));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(u16x2_m, a), __builtin_bit_cast(u16x2_m, b)));
}
__device__ __forceinline__ uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_sub_sat(__builtin_bit_cast(u16x2_m, a), __builtin_bit_cast(u16x2_m, b)));
}

// One row's contribution for the columns x0 .. x0+3: h3 and m2 of the pairs (x0, x0+1) and (x0+2, x0+3), and the pixels
// themselves, every value in a 16-bit half.
struct MinRow { uint32_t h01, h23, m01, m23, c01, c23; };

// ALIGNED (kernel uniform: dword-aligned rows, W % 4 == 0): three dword loads at clamped offsets, unconditional -- a branch
// per row between them made every row's loads wait for the row before (ten dependent round trips per thread: 50 us of
// the kernel's 53) -- and the three groups that need other bytes (the row's first, its last, any beyond its end) patch
// them in afterwards.
struct MinRaw { uint32_t L, M, R; };      // the dwords left of, at and right of x0: bytes b0 = L[3], b1..b4 = M, b5 = R[0]

template <bool ALIGNED>
__device__ __forceinline__ MinRaw minima_load(const uint8_t *row, int W, int x0) {
  MinRaw v;
  if (ALIGNED) {
    const uint32_t last = (uint32_t)(W - 4), o = min((uint32_t)x0, last);
    v.L = *reinterpret_cast<const uint32_t *>(row + (o >= 4u ? o - 4u : 0u));
    v.M = *reinterpret_cast<const uint32_t *>(row + o);
    v.R = *reinterpret_cast<const uint32_t *>(row + min(o + 4u, last));
  } else {
    auto at = [&](int x) { return (uint32_t)row[min(max(x, 0), W - 1)]; };
    v.L = at(x0 - 1) << 24;
    v.M = at(x0) | (at(x0 + 1) << 8) | (at(x0 + 2) << 16) | (at(x0 + 3) << 24);
    v.R = at(x0 + 4);
  }
  return v;
}

template <bool ALIGNED>
__device__ __forceinline__ MinRow minima_row(MinRaw v, int W, int x0) {
  uint32_t L = v.L, M = v.M, R = v.R;
  if (ALIGNED) {
    if (x0 == 0) L = M << 24;                                // the pixel left of the row's first: itself
    if ((uint32_t)x0 >= (uint32_t)(W - 4)) R = M >> 24;      // ... right of its last: itself
    if (x0 >= W) { L = 0u; M = 0u; R = 0u; }                 // a group beyond the row: all equal, no maximum
  }
  // pairs of neighbouring bytes, one per 16-bit half (v_perm_b32: selector 0-3 = bytes of the second operand, 4-7 = of the first, 0x0c = zero)
  const uint32_t A = __builtin_amdgcn_perm(L, M, 0x0C000C07u);      // (b0, b1)
  const uint32_t B = __builtin_amdgcn_perm(M, M, 0x0C010C00u);      // (b1, b2): the pixels x0, x0+1
  const uint32_t C = __builtin_amdgcn_perm(M, M, 0x0C020C01u);      // (b2, b3)
  const uint32_t D = __builtin_amdgcn_perm(M, M, 0x0C030C02u);      // (b3, b4): the pixels x0+2, x0+3
  const uint32_t E = __builtin_amdgcn_perm(R, M, 0x0C040C03u);      // (b4, b5)
  MinRow r;
  r.m01 = pk_max(A, C);
  r.m23 = pk_max(C, E);
  r.h01 = pk_max(r.m01, B);
  r.h23 = pk_max(r.m23, D);
  r.c01 = B;
  r.c23 = D;
  return r;
}

// bit k: pixel x0 + k of row `mid` is a strict maximum of its 3 x 3 window
__device__ __forceinline__ uint32_t minima_nibble(const MinRow &above, const MinRow &mid, const MinRow &below) {
  const uint32_t n01 = pk_max(pk_max(above.h01, below.h01), mid.m01), n23 = pk_max(pk_max(above.h23, below.h23), mid.m23);
  const uint32_t g01 = pk_min(pk_sub_sat(mid.c01, n01), 0x00010001u), g23 = pk_min(pk_sub_sat(mid.c23, n23), 0x00010001u);      // 1 where centre > neighbours
  const uint32_t t = g01 | (g23 << 2);      // bits 0, 16, 2, 18
  return (t | (t >> 15)) & 0xFu;
}

// A workgroup: MIN_RS rows of one 1024-pixel segment (a row strip per workgroup, all segments in turn: 16 waves per CU in
// flight and 73 -> 53 us; this way 8192 workgroups at 8192^2).  Row counts are added up in counts[y], strip counts in
// counts[H + strip]: eight atomics per address.
template <bool ALIGNED>
__global__ __launch_bounds__(256) void k_minima_count(const uint8_t *__restrict__ img, size_t stride, int H, int W,
                                                      uint32_t *__restrict__ counts, uint8_t *__restrict__ nibbles) {
  __shared__ uint32_t s_wave[4 * MIN_RS];
  const int segs = (W + SEG - 1) / SEG;
  const int seg = blockIdx.x % segs, y0 = (blockIdx.x / segs) * MIN_RS;
  auto row_ptr = [&](int y) { return img + (size_t)min(max(y, 0), H - 1) * stride; };      // (workgroup uniform)
  const int x0 = seg * SEG + threadIdx.x * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every load of the strip first (thirty dwords in flight per thread), then the arithmetic, then the stores
  MinRaw raw[MIN_RS + 2];
#pragma unroll
  for (int r = 0; r < MIN_RS + 2; ++r) raw[r] = minima_load<ALIGNED>(row_ptr(y0 + r - 1), W, x0);
  MinRow above = minima_row<ALIGNED>(raw[0], W, x0), mid = minima_row<ALIGNED>(raw[1], W, x0);
  uint32_t mine = 0;      // lane r ends up with the wave's count of row y0 + r
  uint32_t nib[MIN_RS];
#pragma unroll
  for (int r = 0; r < MIN_RS; ++r) {
    const MinRow below = minima_row<ALIGNED>(raw[r + 2], W, x0);
    const uint32_t m = minima_nibble(above, mid, below);
    nib[r] = m;
    const uint32_t c = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((m & 1u) != 0u)) + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((m & 2u) != 0u)) +
                       (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((m & 4u) != 0u)) + (uint32_t)__popcll(__builtin_amdgcn_ballot_w64((m & 8u) != 0u));      // (scalar: the wave's count)
    if (lane == r) mine = c;
    above = mid;
    mid = below;
  }
#pragma unroll
  for (int r = 0; r < MIN_RS; ++r)      // (a group beyond the row's end: 0)
    if (y0 + r < H) nibbles[((size_t)(y0 + r) * segs + seg) * 256 + threadIdx.x] = (uint8_t)nib[r];
  if (lane < MIN_RS) s_wave[wave * MIN_RS + lane] = mine;
  __syncthreads();
  if (threadIdx.x < MIN_RS) {
    const uint32_t c = y0 + (int)threadIdx.x < H ? s_wave[threadIdx.x] + s_wave[MIN_RS + threadIdx.x] + s_wave[2 * MIN_RS + threadIdx.x] + s_wave[3 * MIN_RS + threadIdx.x] : 0u;
    if (c) atomicAdd(&counts[y0 + threadIdx.x], c);
    uint32_t strip = c;      // (MIN_RS == 8 lanes: three steps)
#pragma unroll
    for (int off = MIN_RS / 2; off > 0; off >>= 1) strip += __shfl_down(strip, off, 64);
    if (threadIdx.x == 0 && strip) atomicAdd(&counts[H + y0 / MIN_RS], strip);
  }
}

// counts: minima_segments(h, w) words, zeroed here: [y] row y's count, [h + k] the count of rows 8k .. 8k+7 (minima_write
// turns them into list positions itself: a scan launch of their own cost 8 us, a "last workgroup scans" counter that all
// 8192 workgroups add to 330 -- atomics to ONE address run at ~25 M/s on this chip)
hipError_t minima_count(hipStream_t s, const uint8_t *img, size_t stride, int h, int w, uint32_t *counts, uint8_t *nibbles) {
  if (h == 0 || w == 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(counts, 0, minima_segments(h, w) * sizeof(uint32_t), s);
  if (e != hipSuccess) return e;
  const int segs = (w + SEG - 1) / SEG;
  const unsigned grid = (unsigned)(((h + MIN_RS - 1) / MIN_RS) * segs);
  if (((reinterpret_cast<uintptr_t>(img) | stride) & 3u) == 0 && (w & 3) == 0 && w >= 4)
    k_minima_count<true><<<grid, 256, 0, s>>>(img, stride, h, w, counts, nibbles);
  else
    k_minima_count<false><<<grid, 256, 0, s>>>(img, stride, h, w, counts, nibbles);
  return hipGetLastError();
}

// The write pass: one WAVE per row, no barrier.  A lane takes the nibbles of 32 consecutive pixels (8 bytes) per step, 64 such
// groups a step; a wave scan of their counts gives every group its place in the list.
// mask / word_base (optional, W % 32 == 0): the seed side tables of this very list (k_seed_tables builds them from a list
// it has to search and check; here they fall out of the compaction: a group IS a mask word, its place in the list the
// word's base).  With them the kernel also zeroes the two small arrays that k_seed_tables zeroes for the transform.
// out_rc may then be null: a transform seeded by the image's own minima needs no list.
template <bool LIST>
__global__ __launch_bounds__(256) void k_minima_write(const uint8_t *__restrict__ nibbles, int H, int W,
                                                      const uint32_t *__restrict__ counts, uint32_t *out_rc, size_t cap,
                                                      uint32_t *mask, uint32_t *word_base,
                                                      uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b, uint32_t *total) {
  __shared__ uint32_t s_part[4];
  __shared__ uint2 s_stage[LIST ? 4 : 1][LIST ? 1024 : 1];      // (32 KiB: a step's list entries per wave; none for the tables alone)
  {
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
    for (size_t i = tid; i < n_zero_a; i += step) zero_a[i] = 0u;
    for (size_t i = tid; i < n_zero_b; i += step) zero_b[i] = 0u;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, y0 = blockIdx.x * 4, y = y0 + wave;
  // Where the workgroup's first row starts in the list: the strips before its own (counts[H + k]: at most H / 8 words,
  // L2-resident, summed by all 256 threads), then the rows of its own strip before it.
  const int strip0 = y0 / MIN_RS;
  uint32_t part = 0;
  for (int k = (int)threadIdx.x; k < strip0; k += 256) part += counts[H + k];
  if ((int)threadIdx.x < y0 - strip0 * MIN_RS) part += counts[strip0 * MIN_RS + threadIdx.x];
  for (int off = 32; off > 0; off >>= 1) part += __shfl_down(part, off, 64);
  if (lane == 0) s_part[wave] = part;
  __syncthreads();
  if (y >= H) return;
  size_t pos = (size_t)s_part[0] + s_part[1] + s_part[2] + s_part[3];
  for (int k = 0; k < wave; ++k) pos += counts[y0 + k];
  if (y == H - 1 && lane == 0) *total = (uint32_t)pos + counts[y];
  const int segs = (W + SEG - 1) / SEG, groups = segs * 32;      // 8-byte groups of the row's nibble bytes
  const uint2 *nb = reinterpret_cast<const uint2 *>(nibbles + (size_t)y * segs * 256);
  for (int g0 = 0; g0 < groups; g0 += 64) {
    const int g = g0 + lane;
    const uint2 eight = g < groups ? nb[g] : make_uint2(0u, 0u);
    // eight low nibbles -> one word: bit 4k + j = pixel 32 g + 4k + j
    uint32_t lo = eight.x & 0x0F0F0F0Fu, hi = eight.y & 0x0F0F0F0Fu;
    lo = (lo | (lo >> 4)) & 0x00FF00FFu; hi = (hi | (hi >> 4)) & 0x00FF00FFu;
    lo = (lo | (lo >> 8)) & 0xFFFFu; hi = (hi | (hi >> 8)) & 0xFFFFu;
    uint32_t word = lo | (hi << 16);
    const uint32_t c = __popc(word);
    const uint32_t incl = wave_inclusive_sum(c);
    size_t at = pos + incl - c;
    if (mask && g < groups && g * 32 < W) {      // (W % 32 == 0: the word lies in this row, and in the plane)
      const size_t wi = ((size_t)y * W >> 5) + g;
      mask[wi] = word;
      word_base[wi] = (uint32_t)at;
    }
    const uint32_t step_total = (uint32_t)__shfl((int)incl, 63, 64);
    if (LIST) {
      // the step's entries through LDS: a lane's own are three or four, 28 bytes from its neighbour's -- written where they
      // arise, every store instruction touched dozens of lines; staged, the wave writes them 512 contiguous bytes a time
      // (at most one maximum per 2 x 2 block: 16 a lane, 1024 a step)
      uint2 *stage = s_stage[threadIdx.x >> 6];
      uint32_t k = incl - c;
      while (word) {
        const int b = __builtin_ctz(word);
        word &= word - 1u;
        stage[k++] = make_uint2((uint32_t)y, (uint32_t)(g * 32 + b));
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      for (uint32_t i = (uint32_t)lane; i < step_total; i += 64u)
        if (pos + i < cap) *reinterpret_cast<uint2 *>(out_rc + 2 * (pos + i)) = stage[i];
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");      // the stage is read before the next step fills it
    }
    pos += step_total;
  }
}

hipError_t minima_write(hipStream_t s, const uint8_t *nibbles, int h, int w, const uint32_t *counts, uint32_t *total, uint32_t *out_rc, size_t cap,
                        uint32_t *mask, uint32_t *word_base, uint32_t *zero_a, size_t n_zero_a, uint32_t *zero_b, size_t n_zero_b) {
  if (h == 0 || w == 0) return hipMemsetAsync(total, 0, sizeof(uint32_t), s);
  if (out_rc) k_minima_write<true><<<(h + 3) / 4, 256, 0, s>>>(nibbles, h, w, counts, out_rc, cap, mask, word_base, zero_a, n_zero_a, zero_b, n_zero_b, total);
  else k_minima_write<false><<<(h + 3) / 4, 256, 0, s>>>(nibbles, h, w, counts, out_rc, cap, mask, word_base, zero_a, n_zero_a, zero_b, n_zero_b, total);
  return hipGetLastError();
}

}  // namespace wsk
