// ws_merge.hip -- gfx950 kernels of the merging transform (see ws_merge.hpp for the maths).
//
// Union-find over seed colours, lock-free, union-by-min-index: a root only ever gets hooked
// under a SMALLER index (atomicCAS on the root's own slot) and path halving only ever
// lowers a parent (atomicMin), so every parent chain is monotone and a stale read is still
// an ancestor.  All parent updates are device-scope atomics: XCD L2s are not coherent with
// each other, and nothing here relies on a plain store being seen inside a launch.
#include "ws_common.hpp"
#include "ws_merge.hpp"

#include <algorithm>

namespace wsk {

// Reads of the forest may be stale: parents only ever decrease along a chain, so an old value is
// still an ancestor and the CAS that hooks a root returns the true state.  That makes an ordinary
// L1-cached load legal here (workgroup scope = a plain global_load the compiler will not hoist);
// the agent-scope form bypasses L1 and made every find() of the one surviving root an L2 round trip.
__device__ __forceinline__ uint32_t ld_parent(const uint32_t *parent, uint32_t x) {
  return __hip_atomic_load(parent + x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ uint32_t uf_find(uint32_t *parent, uint32_t x) {
  for (;;) {
    const uint32_t p = ld_parent(parent, x);
    if (p == x) return x;
    const uint32_t g = ld_parent(parent, p);
    if (g == p) return p;
    atomicMin(parent + x, g);          // path halving; parents only decrease
    x = g;
  }
}

// root of x by a plain walk: no halving, no atomics (for a forest nobody is changing during the launch, or a crowd of
// readers of the same few chains, whose halving atomics would queue up on the same words)
__device__ __forceinline__ uint32_t uf_root(const uint32_t *parent, uint32_t x) {
  for (;;) {
    const uint32_t p = ld_parent(parent, x);
    if (p == x) return x;
    x = p;
  }
}

// returns the node that lost its root status (hooked under a smaller root), or 0xFFFFFFFF
__device__ __forceinline__ uint32_t uf_union(uint32_t *parent, uint32_t a, uint32_t b) {
  for (;;) {
    a = uf_find(parent, a);
    b = uf_find(parent, b);
    if (a == b) return 0xFFFFFFFFu;
    if (a > b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicCAS(parent + b, b, a);     // b is a root only while parent[b] == b
    if (old == b) return b;
    b = old;                                              // someone hooked b first: continue from there
  }
}

__global__ void k_uf_init(uint32_t *parent, uint32_t *size, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) { parent[i] = (uint32_t)i; size[i] = 0u; }
}

hipError_t uf_init(hipStream_t s, uint32_t *parent, uint32_t *size, size_t n) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_uf_init<<<blocks, 256, 0, s>>>(parent, size, n);
  return hipGetLastError();
}

// ---- edge / pixel enumeration shared by the histogram, scatter and whole-image union ------

constexpr int MSEG = 1024;   // pixels per workgroup: one row segment, 4 per thread

struct PixelItems {
  // up to 4 pixels of one row segment and their right / down crossings
  uint32_t px_lvl[4], px_col[4];      // level 0xFFFFFFFF = no item
  uint32_t er_lvl[4], ed_lvl[4];
  uint2 er[4], ed[4];
};

__device__ __forceinline__ bool interior(int y, int x, int H, int W) {
  return y >= 1 && y < H - 1 && x >= 1 && x < W - 1;
}

// Loads first, unconditional and on clamped addresses (a load under a data-dependent branch is waited for before the next is
// issued: 24 dependent round trips per thread in the first form of this function), predicates afterwards.  With W % 4 == 0 the
// four pixels and the four below them are 16-byte loads.
typedef uint32_t u32x4_m __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void gather_items(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                             int H, int W, int y, int x0, PixelItems &it) {
  uint32_t kp[5], lp[5], kd[4], ld[4];      // the row's pixels x0 .. x0 + 4 and the pixels below x0 .. x0 + 3
  const int yd = min(y + 1, H - 1);
  if ((W & 3) == 0 && x0 + 3 < W) {
    const size_t p = (size_t)y * W + x0, q = (size_t)yd * W + x0;
    const u32x4_m a = *reinterpret_cast<const u32x4_m *>(keys + p), b = *reinterpret_cast<const u32x4_m *>(labels + p);
    const u32x4_m c = *reinterpret_cast<const u32x4_m *>(keys + q), d = *reinterpret_cast<const u32x4_m *>(labels + q);
    const size_t pr = (size_t)y * W + min(x0 + 4, W - 1);
    kp[4] = keys[pr]; lp[4] = labels[pr];
    kp[0] = a.x; kp[1] = a.y; kp[2] = a.z; kp[3] = a.w;
    lp[0] = b.x; lp[1] = b.y; lp[2] = b.z; lp[3] = b.w;
    kd[0] = c.x; kd[1] = c.y; kd[2] = c.z; kd[3] = c.w;
    ld[0] = d.x; ld[1] = d.y; ld[2] = d.z; ld[3] = d.w;
  } else {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const size_t p = (size_t)y * W + min(x0 + k, W - 1);
      kp[k] = keys[p]; lp[k] = labels[p];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t q = (size_t)yd * W + min(x0 + k, W - 1);
      kd[k] = keys[q]; ld[k] = labels[q];
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    it.px_lvl[k] = it.er_lvl[k] = it.ed_lvl[k] = 0xFFFFFFFFu;
    const int x = x0 + k;
    const bool here = x < W && kp[k] != KEY_INF;            // uncoloured pixels carry no lake
    const uint32_t vp = kp[k] >> 24;
    if (here) { it.px_lvl[k] = vp; it.px_col[k] = lp[k]; }
    const bool ip = interior(y, x, H, W);
    // find_merge only sees pairs around a 3x3 window centre (lib.rs:411-434)
    if (here && x + 1 < W && kp[k + 1] != KEY_INF && lp[k + 1] != lp[k] && (ip || interior(y, x + 1, H, W))) {
      it.er_lvl[k] = max(vp, kp[k + 1] >> 24);
      it.er[k] = make_uint2(lp[k], lp[k + 1]);
    }
    if (here && y + 1 < H && kd[k] != KEY_INF && ld[k] != lp[k] && (ip || interior(y + 1, x, H, W))) {
      it.ed_lvl[k] = max(vp, kd[k] >> 24);
      it.ed[k] = make_uint2(lp[k], ld[k]);
    }
  }
}

// A workgroup takes SEG_RUN consecutive row segments of a large plane (one per step), not one: the 512 level counters are hit once per
// workgroup and level, and with a workgroup per 1024 pixels those same-address atomics (~12 ns each) WERE the kernel --
// 8192^2: 33 M of them on 512 words, 1.4 ms for 0.5 GB of reads.
constexpr int SEG_RUN_MAX = 16;
static int seg_run_for(size_t total) { return (int)std::min<size_t>(std::max<size_t>(total / 4096, 1), SEG_RUN_MAX); }      // small planes keep a workgroup per segment

__global__ __launch_bounds__(256) void k_level_hist(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                                    int H, int W, int segs, int SEG_RUN, u64c *hist_px, u64c *hist_edge) {
  __shared__ uint32_t s_px[NLEVELS], s_ed[NLEVELS];
  s_px[threadIdx.x] = 0;
  s_ed[threadIdx.x] = 0;
  __syncthreads();
  const size_t total = (size_t)H * segs;
  for (int j = 0; j < SEG_RUN; ++j) {
    const size_t sg = (size_t)blockIdx.x * SEG_RUN + j;
    if (sg >= total) break;
    const int y = (int)(sg / segs), seg = (int)(sg % segs);
    PixelItems it;
    gather_items(keys, labels, H, W, y, seg * MSEG + threadIdx.x * 4, it);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (it.px_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_px[it.px_lvl[k]], 1u);
      if (it.er_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.er_lvl[k]], 1u);
      if (it.ed_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.ed_lvl[k]], 1u);
    }
  }
  __syncthreads();
  if (s_px[threadIdx.x]) atomicAdd(&hist_px[threadIdx.x], (u64c)s_px[threadIdx.x]);
  if (s_ed[threadIdx.x]) atomicAdd(&hist_edge[threadIdx.x], (u64c)s_ed[threadIdx.x]);
}

hipError_t level_hist(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                      u64c *hist_px, u64c *hist_edge) {
  if (h == 0 || w == 0) return hipSuccess;
  const int segs = (w + MSEG - 1) / MSEG;
  const size_t total = (size_t)h * segs;
  const int run = seg_run_for(total);
  k_level_hist<<<(unsigned)((total + run - 1) / run), 256, 0, s>>>(keys, labels, h, w, segs, run, hist_px, hist_edge);
  return hipGetLastError();
}

// Exclusive prefix sums of the two histograms, on the device: off[l] = first item of level l, off[256] = total; the
// scatter cursors start at the same values.  (The host used to do this between two synchronisations.)
__global__ __launch_bounds__(NLEVELS) void k_level_offsets(const u64c *__restrict__ hist_px, const u64c *__restrict__ hist_ed,
                                                           u64c *off_px, u64c *off_ed, u64c *cur_px, u64c *cur_ed) {
  __shared__ u64c s_a[NLEVELS], s_b[NLEVELS];
  const int t = threadIdx.x;
  const u64c a0 = hist_px[t], b0 = hist_ed[t];
  s_a[t] = a0;
  s_b[t] = b0;
  __syncthreads();
  for (int o = 1; o < NLEVELS; o <<= 1) {            // Hillis-Steele inclusive scan
    const u64c a = t >= o ? s_a[t - o] : 0ull, b = t >= o ? s_b[t - o] : 0ull;
    __syncthreads();
    s_a[t] += a;
    s_b[t] += b;
    __syncthreads();
  }
  off_px[t] = cur_px[t] = s_a[t] - a0;
  off_ed[t] = cur_ed[t] = s_b[t] - b0;
  if (t == NLEVELS - 1) { off_px[NLEVELS] = s_a[t]; off_ed[NLEVELS] = s_b[t]; }
}

hipError_t level_offsets(hipStream_t s, const u64c *hist_px, const u64c *hist_ed, u64c *off_px, u64c *off_ed, u64c *cur_px, u64c *cur_ed) {
  k_level_offsets<<<1, NLEVELS, 0, s>>>(hist_px, hist_ed, off_px, off_ed, cur_px, cur_ed);
  return hipGetLastError();
}

// Two walks over the workgroup's SEG_RUN segments: the first counts its items per level in LDS, then ONE reservation per
// (workgroup, level), then the second walk gathers the items again (L2 still holds them) and writes them behind LDS cursors.
__global__ __launch_bounds__(256) void k_level_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                                       int H, int W, int segs, int SEG_RUN, u64c *cursor_px, u64c *cursor_edge,
                                                       uint32_t *px_items, uint2 *edge_items) {
  __shared__ uint32_t s_px[NLEVELS], s_ed[NLEVELS];
  __shared__ u64c s_bpx[NLEVELS], s_bed[NLEVELS];
  s_px[threadIdx.x] = 0;
  s_ed[threadIdx.x] = 0;
  __syncthreads();
  const size_t total = (size_t)H * segs;
  for (int j = 0; j < SEG_RUN; ++j) {
    const size_t sg = (size_t)blockIdx.x * SEG_RUN + j;
    if (sg >= total) break;
    PixelItems it;
    gather_items(keys, labels, H, W, (int)(sg / segs), (int)(sg % segs) * MSEG + threadIdx.x * 4, it);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (it.px_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_px[it.px_lvl[k]], 1u);
      if (it.er_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.er_lvl[k]], 1u);
      if (it.ed_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.ed_lvl[k]], 1u);
    }
  }
  __syncthreads();
  // one global reservation per (workgroup, level) that has items
  {
    const uint32_t cp = s_px[threadIdx.x], ce = s_ed[threadIdx.x];
    s_bpx[threadIdx.x] = cp ? atomicAdd(&cursor_px[threadIdx.x], (u64c)cp) : 0ull;
    s_bed[threadIdx.x] = ce ? atomicAdd(&cursor_edge[threadIdx.x], (u64c)ce) : 0ull;
  }
  __syncthreads();
  s_px[threadIdx.x] = 0;
  s_ed[threadIdx.x] = 0;
  __syncthreads();
  for (int j = 0; j < SEG_RUN; ++j) {
    const size_t sg = (size_t)blockIdx.x * SEG_RUN + j;
    if (sg >= total) break;
    PixelItems it;
    gather_items(keys, labels, H, W, (int)(sg / segs), (int)(sg % segs) * MSEG + threadIdx.x * 4, it);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (it.px_lvl[k] != 0xFFFFFFFFu) {
        const uint32_t l = it.px_lvl[k];
        px_items[s_bpx[l] + atomicAdd(&s_px[l], 1u)] = it.px_col[k];
      }
      if (it.er_lvl[k] != 0xFFFFFFFFu) {
        const uint32_t l = it.er_lvl[k];
        edge_items[s_bed[l] + atomicAdd(&s_ed[l], 1u)] = it.er[k];
      }
      if (it.ed_lvl[k] != 0xFFFFFFFFu) {
        const uint32_t l = it.ed_lvl[k];
        edge_items[s_bed[l] + atomicAdd(&s_ed[l], 1u)] = it.ed[k];
      }
    }
  }
}

hipError_t level_scatter(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                         u64c *cursor_px, u64c *cursor_edge, uint32_t *px_items, uint2 *edge_items) {
  if (h == 0 || w == 0) return hipSuccess;
  const int segs = (w + MSEG - 1) / MSEG;
  const size_t total = (size_t)h * segs;
  const int run = seg_run_for(total);
  k_level_scatter<<<(unsigned)((total + run - 1) / run), 256, 0, s>>>(keys, labels, h, w, segs, run, cursor_px, cursor_edge, px_items, edge_items);
  return hipGetLastError();
}

// ---- per-level union / sizes / emit -----------------------------------------------------

// range (nullable): {first, end} item indices in device memory -- the level's bucket, whose bounds the host never reads
// death (nullable): death[c] = the level at which colour c stopped being a root (0xFFFFFFFF: still one)
__device__ __forceinline__ void union_body(const uint2 *__restrict__ edges, size_t n, uint32_t *parent, uint32_t *hooked,
                                           uint32_t *hooked_count, const u64c *__restrict__ range, uint32_t *death, uint32_t level,
                                           unsigned nblocks) {
  if (range) { edges += range[0]; n = (size_t)(range[1] - range[0]); }
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)nblocks * blockDim.x;
  for (; i < n; i += step) {
    const uint2 e = edges[i];
    const uint32_t lost = uf_union(parent, e.x, e.y);
    if (lost != 0xFFFFFFFFu) {
      if (hooked) hooked[atomicAdd(hooked_count, 1u)] = lost;
      if (death) death[lost] = level;
    }
  }
}

__global__ void k_union_edges(const uint2 *__restrict__ edges, size_t n, uint32_t *parent, uint32_t *hooked,
                              uint32_t *hooked_count, const u64c *__restrict__ range) {
  union_body(edges, n, parent, hooked, hooked_count, range, nullptr, 0u, gridDim.x);
}

hipError_t union_edges(hipStream_t s, const uint2 *edges, size_t n, uint32_t *parent, uint32_t *hooked,
                       uint32_t *hooked_count) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  k_union_edges<<<blocks, 256, 0, s>>>(edges, n, parent, hooked, hooked_count, nullptr);
  return hipGetLastError();
}

hipError_t union_edges_ranged(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned grid, uint32_t *parent, uint32_t *hooked,
                              uint32_t *hooked_count) {
  k_union_edges<<<grid, 256, 0, s>>>(edge_items, 0, parent, hooked, hooked_count, range);
  return hipGetLastError();
}

// After a level's unions every node hooked in this level hands its accumulated area to its final root, and the areas
// of the pixels arriving at this level go to the lake they arrive in.  Late levels have few lakes, so most lanes of a
// wave add to the SAME word: the wave adds once per distinct root (leader election by ballot) -- same-address atomics
// retire one per ~12 ns, and 4000 of them were the whole kernel.
// Both in one launch (a level of transform_to_list was four dependent launches of a few microseconds each:
// launch gaps are most of its time).  They do not interfere: folding reads the areas of nodes hooked in this level --
// no longer roots, so no arrival is added to them -- and both add to roots.
__global__ void k_fold_and_add(const uint32_t *__restrict__ hooked, const uint32_t *__restrict__ hooked_count,
                               const uint32_t *__restrict__ px_items, size_t n, uint32_t *parent, uint32_t *size,
                               const u64c *__restrict__ range, int split) {
  if (range) { px_items += range[0]; n = (size_t)(range[1] - range[0]); }
  const int lane = threadIdx.x & 63;
  // the upper half of the grid folds, the lower half counts arrivals (`split`): two chains of dependent L2 round trips
  // side by side instead of one after the other
  const unsigned half = split ? gridDim.x / 2 : gridDim.x;
  const bool folds = !split || blockIdx.x >= half, counts = !split || blockIdx.x < half;
  const unsigned bid = split && blockIdx.x >= half ? blockIdx.x - half : blockIdx.x;
  const size_t step = (size_t)half * blockDim.x;
  if (hooked && folds) {
    const uint32_t nh = *hooked_count;
    for (size_t i = (size_t)bid * blockDim.x + threadIdx.x; i < nh; i += step) {
      const uint32_t b = hooked[i];
      const uint32_t area = size[b];
      if (area) atomicAdd(&size[uf_find(parent, b)], area);
    }
  }
  if (!counts) return;
  for (size_t base = (size_t)bid * blockDim.x; base < n; base += step) {      // uniform trip count per wave
    const size_t i = base + threadIdx.x;
    const bool active = i < n;
    const uint32_t r = active ? uf_find(parent, px_items[i]) : 0xFFFFFFFFu;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
    // a few rounds of leader election catch the case that matters (late levels: a handful of lakes, thousands of lanes on
    // the same word); what is left after them -- early levels: every lane another lake -- adds for itself
    for (int round = 0; round < 4 && todo != 0; ++round) {
      const int leader = (int)__builtin_ctzll(todo);
      const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);      // (v_readlane: no LDS round trip; leader is wave uniform)
      const unsigned long long same = __builtin_amdgcn_ballot_w64(active && r == r0);
      if (lane == leader) atomicAdd(&size[r0], (uint32_t)__popcll(same));
      todo &= ~same;
    }
    if ((todo >> lane) & 1ull) atomicAdd(&size[r], 1u);
  }
}

hipError_t fold_and_add(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items, size_t n,
                        uint32_t *parent, uint32_t *size) {
  if (!hooked && n == 0) return hipSuccess;
  const size_t want = std::max<size_t>((n + 255) / 256, hooked ? 512 : 1);
  k_fold_and_add<<<(unsigned)std::min<size_t>(want, 4096), 256, 0, s>>>(hooked, hooked_count, px_items, n, parent, size, nullptr, 0);
  return hipGetLastError();
}

hipError_t fold_and_add_ranged(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items,
                               const u64c *range, unsigned grid, uint32_t *parent, uint32_t *size) {
  k_fold_and_add<<<hooked ? 2 * grid : grid, 256, 0, s>>>(hooked, hooked_count, px_items, 0, parent, size, range, hooked ? 1 : 0);
  return hipGetLastError();
}

// lib.rs:628-635 sparsely: one (colour, area) record per lake with area > 0.  Records of one level are contiguous:
// a record's position is the number of records of the earlier levels -- every workgroup adds up their counters, which
// earlier launches finished -- plus a ticket from this level's own counter, one request per WAVE (its lakes take
// consecutive records).  No cursor is handed from level to level inside the kernel: the "last workgroup publishes the
// total" protocol that did that needs __threadfence(), which on this part writes the XCD's L2 back -- 24 us per level
// for a kernel with 5 us of work, 6 of the 11 ms of a 1024^2 transform_to_list.
constexpr int EMIT_PER_THREAD = 4;      // colours per thread: a workgroup of 256 looks at 1024 colours and asks for ONE ticket range
// death (nullable): a colour is a lake of `level` while death[c] > level -- the test that stays true while the NEXT level's
// unions are already hooking roots (k_union_emit); without it: parent[c] == c.
__device__ __forceinline__ void emit_body(const uint32_t *__restrict__ parent, const uint32_t *__restrict__ size, size_t n_colours,
                                          uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t level,
                                          const uint32_t *__restrict__ death, unsigned bid) {
  __shared__ u64c s_base;
  __shared__ uint32_t s_wave[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) {
    u64c before = 0;
    for (uint32_t j = (uint32_t)lane; j < level; j += 64) before += level_counts[j];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) s_base = before;
  }
  // same-address atomics retire one per ~10 ns: a ticket request per wave (1.8 k per level at 1024^2) was 17 us of this
  // kernel; per workgroup of 1024 colours it is ~1 us
  const size_t i0 = (size_t)bid * (256 * EMIT_PER_THREAD) + 1;      // colour 0 = uncoloured
  uint32_t area[EMIT_PER_THREAD];
  bool lake[EMIT_PER_THREAD];
  unsigned long long m[EMIT_PER_THREAD];
  uint32_t mine = 0;
#pragma unroll
  for (int k = 0; k < EMIT_PER_THREAD; ++k) {
    const size_t i = i0 + (size_t)k * 256 + threadIdx.x;
    area[k] = i < n_colours ? size[i] : 0u;
    lake[k] = i < n_colours && area[k] != 0u && (death ? death[i] > level : parent[i] == (uint32_t)i);
    m[k] = __builtin_amdgcn_ballot_w64(lake[k]);
    mine += (uint32_t)__popcll(m[k]);
  }
  if (lane == 0) s_wave[wave] = mine;
  __syncthreads();
  const uint32_t w0 = s_wave[0], w1 = s_wave[1], w2 = s_wave[2], w3 = s_wave[3];
  const uint32_t block_total = w0 + w1 + w2 + w3;
  if (block_total == 0) return;            // workgroup uniform
  __shared__ u64c s_first;
  if (threadIdx.x == 0) s_first = atomicAdd(level_counts + level, (u64c)block_total);
  __syncthreads();
  u64c pos = s_base + s_first + (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u);
#pragma unroll
  for (int k = 0; k < EMIT_PER_THREAD; ++k) {
    const u64c p = pos + (u64c)__popcll(m[k] & ((1ull << lane) - 1ull));
    if (lake[k] && p < cap) {
      const size_t i = i0 + (size_t)k * 256 + threadIdx.x;
      lakes[2 * p] = (uint64_t)i;
      lakes[2 * p + 1] = (uint64_t)area[k];
    }
    pos += (u64c)__popcll(m[k]);
  }
}

__global__ __launch_bounds__(256) void k_emit_lakes(const uint32_t *__restrict__ parent, const uint32_t *__restrict__ size, size_t n_colours,
                                                    uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t level,
                                                    const uint32_t *__restrict__ death) {
  emit_body(parent, size, n_colours, lakes, cap, level_counts, level, death, blockIdx.x);
}

// The unions of level `level` and the lake records of level `level - 1` in ONE launch (merging transform_to_list: two
// launches per level instead of three).  They do not interfere: the records read the areas (`size`: only the fold
// kernels write it) and decide "was a lake at level - 1" by death[c] > level - 1, which a union of THIS level -- it sets
// death[c] = level -- leaves true; the unions touch parent / hooked / death only.
__global__ __launch_bounds__(256) void k_union_emit(const uint2 *__restrict__ edge_items, const u64c *__restrict__ range, unsigned union_blocks,
                                                    uint32_t *parent, uint32_t *hooked, uint32_t *hooked_count, uint32_t *death, uint32_t level,
                                                    const uint32_t *__restrict__ size, size_t n_colours, unsigned emit_blocks,
                                                    uint64_t *lakes, size_t cap, u64c *level_counts) {
  // different workgroups for the two jobs: both are chains of dependent L2 round trips (find, find, CAS / load, ticket,
  // store), and one workgroup doing one after the other took the sum of the two latencies (8.8 us per launch)
  if (blockIdx.x < union_blocks) union_body(edge_items, 0, parent, hooked, hooked_count, range, death, level, union_blocks);
  else if (level > 0 && blockIdx.x - union_blocks < emit_blocks)
    emit_body(parent, size, n_colours, lakes, cap, level_counts, level - 1, death, blockIdx.x - union_blocks);
}

static size_t emit_blocks_for(size_t n_colours) {
  const size_t per_block = 256 * EMIT_PER_THREAD;
  return n_colours <= 1 ? 1 : (n_colours - 1 + per_block - 1) / per_block;
}

hipError_t union_emit(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned union_grid, uint32_t *parent, uint32_t *hooked,
                      uint32_t *hooked_count, uint32_t *death, uint32_t level, const uint32_t *size, size_t n_colours, uint64_t *lakes,
                      size_t cap, u64c *level_counts) {
  const unsigned eb = (unsigned)emit_blocks_for(n_colours);
  k_union_emit<<<union_grid + (level > 0 ? eb : 0u), 256, 0, s>>>(edge_items, range, union_grid, parent, hooked, hooked_count, death, level,
                                                                        size, n_colours, eb, lakes, cap, level_counts);
  return hipGetLastError();
}

hipError_t emit_lakes(hipStream_t s, const uint32_t *parent, const uint32_t *size, size_t n_colours,
                      uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t level, const uint32_t *death) {
  k_emit_lakes<<<(unsigned)emit_blocks_for(n_colours), 256, 0, s>>>(parent, size, n_colours, lakes, cap, level_counts, level, death);
  return hipGetLastError();
}

// ---- merging transform_to_list at size: records from the list of LIVE lakes ------------------------------------------------
//
// k_emit_lakes looks at every colour at every level: 255 x S visits (8192^2: 1.9 G) for 0.62 G records -- 45 of the 67 ms of a
// transform_to_list there.  A lake of level l was a lake of level l - 1 (every seed's colour holds its seed pixel from the
// start, lib.rs:1670-1677, and lakes only ever merge), so level l's lakes are found among level l - 1's: the kernel below reads
// the list of the lakes alive at the level before (all colours, for level 0), keeps those that are still roots with pixels, and
// writes their records AND the next list in the same positions (one ticket per 4096 candidates).  Work per level: the live lakes.
// sd[c] = (area, death level): one 8-byte gather per candidate.
constexpr int ALIVE_PER_THREAD = 16;
constexpr int ALIVE_CHUNK = 256 * ALIVE_PER_THREAD;

__global__ void k_sd_init(uint2 *sd, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) sd[i] = make_uint2(0u, 0xFFFFFFFFu);
}

hipError_t sd_init(hipStream_t s, uint2 *sd, size_t n) {
  if (n == 0) return hipSuccess;
  k_sd_init<<<(unsigned)std::min<size_t>((n + 1023) / 1024, 8192), 256, 0, s>>>(sd, n);
  return hipGetLastError();
}

// level L's records (and live list) from level L - 1's live list.  bid / nblocks: this job's share of the launch.
__device__ __forceinline__ void emit_alive_body(const uint2 *__restrict__ sd, size_t n_colours, const uint32_t *__restrict__ alive_in,
                                                uint32_t *alive_out, uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t L,
                                                unsigned bid, unsigned nblocks) {
  __shared__ u64c s_base, s_first;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x < 64) {      // records of the earlier levels: their counters are final (earlier launches)
    u64c before = 0;
    for (uint32_t j = (uint32_t)lane; j < L; j += 64) before += level_counts[j];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if (lane == 0) s_base = before;
  }
  const size_t n_in = L == 0 ? (n_colours > 0 ? n_colours - 1 : 0) : (size_t)level_counts[L - 1];
  // Candidates are dealt out lane by lane (slab k of a chunk = 256 consecutive candidates: a wave's loads and its sd gathers
  // are coalesced while the list is sorted) and the survivors KEEP THEIR ORDER inside the chunk (chunks draw their tickets in
  // about the order of their indices), so the list stays nearly sorted by colour from level to level.  (Written out in
  // (wave, slab, lane) order the list was reshuffled at every level and level 50's gathers -- 4 M live lakes -- took twice as
  // long as level 0's 7.3 M sequential ones; sixteen CONSECUTIVE candidates per thread kept the order but made every load a
  // 64-line gather: 97 us for level 0 instead of 40.)
  __shared__ uint32_t s_cnt[ALIVE_PER_THREAD][4];
  for (size_t chunk = bid; chunk * ALIVE_CHUNK < n_in; chunk += nblocks) {
    uint32_t col[ALIVE_PER_THREAD], area[ALIVE_PER_THREAD];
    unsigned long long m[ALIVE_PER_THREAD];
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k) {
      const size_t i = chunk * ALIVE_CHUNK + (size_t)k * 256 + threadIdx.x;
      col[k] = i < n_in ? (L == 0 ? (uint32_t)i + 1u : alive_in[i]) : 0u;      // colour 0 = uncoloured: never a lake
    }
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k) {
      const uint2 v = sd[col[k]];
      area[k] = v.x;
      m[k] = __builtin_amdgcn_ballot_w64(col[k] != 0u && v.x != 0u && v.y > L);
    }
    __syncthreads();      // (s_cnt / s_first of the previous chunk have been read)
    if (lane < ALIVE_PER_THREAD) {
      unsigned long long mk = m[0];
#pragma unroll
      for (int k = 1; k < ALIVE_PER_THREAD; ++k) mk = lane == k ? m[k] : mk;
      s_cnt[lane][wave] = (uint32_t)__popcll(mk);
    }
    __syncthreads();
    // exclusive prefix over (slab, wave) in that order: where this wave's survivors of slab k start
    uint32_t start[ALIVE_PER_THREAD], run = 0;
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k) {
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w == wave) start[k] = run;
        run += s_cnt[k][w];
      }
    }
    const uint32_t block_total = run;
    if (block_total == 0) continue;            // workgroup uniform
    if (threadIdx.x == 0) s_first = atomicAdd(level_counts + L, (u64c)block_total);
    __syncthreads();
    const u64c first = s_first;
#pragma unroll
    for (int k = 0; k < ALIVE_PER_THREAD; ++k) {
      if ((m[k] >> lane) & 1ull) {
        const u64c r = first + start[k] + (u64c)__popcll(m[k] & ((1ull << lane) - 1ull));      // position inside the level
        alive_out[r] = col[k];
        const u64c p = s_base + r;
        if (p < cap) {
          lakes[2 * p] = (uint64_t)col[k];
          lakes[2 * p + 1] = (uint64_t)area[k];
        }
      }
    }
  }
}

// the unions of `level` (sd[c].y = level for every root they hook) and, side by side, the records of level - 1
__global__ __launch_bounds__(256) void k_union_emit_alive(const uint2 *__restrict__ edge_items, const u64c *__restrict__ range, unsigned union_blocks,
                                                          uint32_t *parent, uint32_t *hooked, uint32_t *hooked_count, uint2 *sd, uint32_t level,
                                                          size_t n_colours, const uint32_t *__restrict__ alive_in, uint32_t *alive_out,
                                                          uint64_t *lakes, size_t cap, u64c *level_counts) {
  if (blockIdx.x < union_blocks) {
    edge_items += range[0];
    const size_t n = (size_t)(range[1] - range[0]);
    const size_t step = (size_t)union_blocks * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
      const uint2 e = edge_items[i];
      const uint32_t lost = uf_union(parent, e.x, e.y);
      if (lost != 0xFFFFFFFFu) {
        hooked[atomicAdd(hooked_count, 1u)] = lost;
        sd[lost].y = level;
      }
    }
  } else if (level > 0) {
    emit_alive_body(sd, n_colours, alive_in, alive_out, lakes, cap, level_counts, level - 1, blockIdx.x - union_blocks, gridDim.x - union_blocks);
  }
}

__global__ __launch_bounds__(256) void k_emit_alive(const uint2 *__restrict__ sd, size_t n_colours, const uint32_t *__restrict__ alive_in,
                                                    uint32_t *alive_out, uint64_t *lakes, size_t cap, u64c *level_counts, uint32_t L) {
  emit_alive_body(sd, n_colours, alive_in, alive_out, lakes, cap, level_counts, L, blockIdx.x, gridDim.x);
}

// alive: two lists of alive_list_words(n_colours) words; level L reads list (L + 1) & 1 and writes list L & 1
size_t alive_list_words(size_t n_colours) { return (n_colours + 3) & ~(size_t)3; }      // 16-byte aligned lists

hipError_t union_emit_alive(hipStream_t s, const uint2 *edge_items, const u64c *range, unsigned union_grid, uint32_t *parent, uint32_t *hooked,
                            uint32_t *hooked_count, uint2 *sd, uint32_t level, size_t n_colours, uint32_t *alive, unsigned emit_grid,
                            uint64_t *lakes, size_t cap, u64c *level_counts) {
  const uint32_t L = level - 1;      // the level whose records ride along (level > 0)
  const size_t stride = alive_list_words(n_colours);
  k_union_emit_alive<<<union_grid + (level > 0 ? emit_grid : 0u), 256, 0, s>>>(edge_items, range, union_grid, parent, hooked, hooked_count, sd, level,
                                                                             n_colours, alive + (size_t)((L + 1) & 1u) * stride,
                                                                             alive + (size_t)(L & 1u) * stride, lakes, cap, level_counts);
  return hipGetLastError();
}

hipError_t emit_alive(hipStream_t s, const uint2 *sd, size_t n_colours, uint32_t *alive, unsigned emit_grid, uint64_t *lakes, size_t cap,
                      u64c *level_counts, uint32_t L) {
  const size_t stride = alive_list_words(n_colours);
  k_emit_alive<<<emit_grid, 256, 0, s>>>(sd, n_colours, alive + (size_t)((L + 1) & 1u) * stride, alive + (size_t)(L & 1u) * stride, lakes, cap,
                                         level_counts, L);
  return hipGetLastError();
}

// k_fold_and_add on sd[].x, for planes where a level brings hundreds of thousands of pixels to a handful of lakes: a wave adds
// up the RUN of items that go to one root (the common case late in the flood: all of them) over its four steps and adds once,
// and the four waves of a workgroup that end on the same root add once together -- a same-address atomic retires every ~12 ns,
// and one per wave and step was 49 of the kernel's 74 us per level at 8192^2.
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {      // total of v over the wave, in every lane
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o, 64);
  return v;
}

constexpr int FOLD_STEPS = 4;      // items per thread of the counting half

__global__ __launch_bounds__(256) void k_fold_and_add_sd(const uint32_t *__restrict__ hooked, const uint32_t *__restrict__ hooked_count,
                                                         const uint32_t *__restrict__ px_items, const u64c *__restrict__ range,
                                                         uint32_t *parent, uint2 *sd) {
  __shared__ uint32_t s_root[4], s_cnt[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned half = gridDim.x / 2;
  uint32_t *size = reinterpret_cast<uint32_t *>(sd);      // area of colour c: size[2 c]
  if (blockIdx.x >= half) {
    // folds: the areas of the roots hooked at this level go to the roots they ended under
    const unsigned bid = blockIdx.x - half;
    const uint32_t nh = *hooked_count;
    for (size_t base = (size_t)bid * 256; base < nh; base += (size_t)half * 256) {
      const size_t i = base + threadIdx.x;
      const bool active = i < nh;
      const uint32_t b = active ? hooked[i] : 0u;
      const uint32_t area = active ? size[2 * (size_t)b] : 0u;
      const uint32_t r = active && area ? uf_find(parent, b) : 0xFFFFFFFFu;
      unsigned long long todo = __builtin_amdgcn_ballot_w64(r != 0xFFFFFFFFu);
      for (int round = 0; round < 2 && todo != 0; ++round) {      // the one or two lakes that swallow most of the others
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);
        const bool same = r == r0;
        const uint32_t sum = wave_sum_u32(same ? area : 0u);
        if (lane == leader) atomicAdd(&size[2 * (size_t)r0], sum);
        todo &= ~__builtin_amdgcn_ballot_w64(same);
      }
      if ((todo >> lane) & 1ull) atomicAdd(&size[2 * (size_t)r], area);
    }
    return;
  }
  // arrivals: a workgroup takes 256 * FOLD_STEPS consecutive items
  const u64c first = range[0];
  const size_t n = (size_t)(range[1] - first);
  px_items += first;
  for (size_t base = (size_t)blockIdx.x * (256 * FOLD_STEPS); base < n; base += (size_t)half * (256 * FOLD_STEPS)) {
    uint32_t run_root = 0xFFFFFFFFu, run_cnt = 0;      // wave uniform
#pragma unroll
    for (int st = 0; st < FOLD_STEPS; ++st) {
      const size_t i = base + (size_t)st * 256 + threadIdx.x;
      const bool active = i < n;
      const uint32_t r = active ? uf_find(parent, px_items[i]) : 0xFFFFFFFFu;
      unsigned long long todo = __builtin_amdgcn_ballot_w64(active);
      if (todo != 0) {
        const int leader = (int)__builtin_ctzll(todo);
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_readlane((int)r, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(active && r == r0);
        if (r0 != run_root) {
          if (run_cnt && lane == 0) atomicAdd(&size[2 * (size_t)run_root], run_cnt);
          run_root = r0;
          run_cnt = 0;
        }
        run_cnt += (uint32_t)__popcll(same);
        todo &= ~same;
        // what is left (early levels: every lane another lake): two more rounds of leader election, then one add per lane
        for (int round = 0; round < 2 && todo != 0; ++round) {
          const int l2 = (int)__builtin_ctzll(todo);
          const uint32_t r2 = (uint32_t)__builtin_amdgcn_readlane((int)r, l2);
          const unsigned long long s2 = __builtin_amdgcn_ballot_w64(active && r == r2);
          if (lane == l2) atomicAdd(&size[2 * (size_t)r2], (uint32_t)__popcll(s2));
          todo &= ~s2;
        }
        if ((todo >> lane) & 1ull) atomicAdd(&size[2 * (size_t)r], 1u);
      }
    }
    // the waves' runs: one add per distinct root of the workgroup (all four equal late in the flood)
    __syncthreads();
    if (lane == 0) { s_root[wave] = run_cnt ? run_root : 0xFFFFFFFFu; s_cnt[wave] = run_cnt; }
    __syncthreads();
    if (threadIdx.x < 4 && s_root[threadIdx.x] != 0xFFFFFFFFu) {
      bool first_of_its_root = true;
      uint32_t total = 0;
      for (int k = 0; k < 4; ++k)
        if (s_root[k] == s_root[threadIdx.x]) { if (k < (int)threadIdx.x) first_of_its_root = false; total += s_cnt[k]; }
      if (first_of_its_root) atomicAdd(&size[2 * (size_t)s_root[threadIdx.x]], total);
    }
  }
}

hipError_t fold_and_add_sd(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, const uint32_t *px_items, const u64c *range,
                           unsigned grid, uint32_t *parent, uint2 *sd) {
  k_fold_and_add_sd<<<2 * grid, 256, 0, s>>>(hooked, hooked_count, px_items, range, parent, sd);
  return hipGetLastError();
}

// ---- merging across the row blocks of a tiled field (third row) -----------------------------------------
//
// Final labels only.  Every rank holds a union-find over ALL seed colours of the field and joins the touching colours of
// its own block: every horizontal pair of its rows and every vertical pair (r, r + 1), seam pairs to its halo rows
// included, under find_merge's rule that one pixel of a pair is interior IN THE WHOLE FIELD (lib.rs:411-434; `row0` is
// the field row of the block's first local row).  A lake that spans several blocks is a chain of such local pieces
// linked at boundary colours, so it is enough that every rank tells every other, for each colour c on its two boundary
// rows and its halo rows, the root of c in ITS forest: k_block_colour_roots writes the pairs (c, root(c)), one
// all-gather, and every rank joins all gathered pairs into its own forest (union_edges).  Roots are class minima
// (union-by-min), so the merged root is the smallest seed colour of the whole lake: the canonical id.
// (col0, W: a tile of a field cut in both directions holds the field's columns [col0, col0 + w); a row block: col0 = 0, W = w)
__global__ __launch_bounds__(256) void k_block_union_pixels(const uint32_t *__restrict__ labels, int h, int w, int row0, int H,
                                                            uint32_t *parent, int col0, int W) {
  const size_t n = (size_t)h * w;
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; p < n; p += step) {
    const uint32_t lp = labels[p];
    if (lp == 0u) continue;                              // at the final level a pixel is coloured iff its label is not 0
    const int y = (int)(p / (size_t)w), x = (int)(p - (size_t)y * w);
    const bool ip = interior(row0 + y, col0 + x, H, W);
    if (x + 1 < w) {
      const uint32_t lq = labels[p + 1];
      if (lq != 0u && lq != lp && (ip || interior(row0 + y, col0 + x + 1, H, W))) (void)uf_union(parent, lp, lq);
    }
    if (y + 1 < h) {
      const uint32_t lq = labels[p + w];
      if (lq != 0u && lq != lp && (ip || interior(row0 + y + 1, col0 + x, H, W))) (void)uf_union(parent, lp, lq);
    }
  }
}

hipError_t block_union_pixels(hipStream_t s, const uint32_t *labels, int h, int w, int row0, int H, uint32_t *parent, int col0, int W) {
  if (h == 0 || w == 0) return hipSuccess;
  const size_t n = (size_t)h * w;
  k_block_union_pixels<<<(unsigned)std::min<size_t>((n + 255) / 256, 8192), 256, 0, s>>>(labels, h, w, row0, H, parent, col0, W < 0 ? w : W);
  return hipGetLastError();
}

// ... and of a tile: its two outermost rows AND columns on every side (4 w + 4 h pairs, then (0, 0) up to n_pairs: tiles of
// a grid differ by a row or a column, an all-gather wants equal parts)
__global__ void k_block_colour_roots2d(const uint32_t *__restrict__ labels, int h, int w, uint32_t *parent, uint2 *pairs, size_t n_pairs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_pairs) return;
  uint32_t c = 0;
  if (i < (size_t)4 * w) {
    const int k = (int)(i / (size_t)w), x = (int)(i - (size_t)k * w);
    const int rows[4] = {0, min(1, h - 1), max(h - 2, 0), h - 1};
    c = labels[(size_t)rows[k] * w + x];
  } else if (i < (size_t)4 * w + (size_t)4 * h) {
    const size_t j = i - (size_t)4 * w;
    const int k = (int)(j / (size_t)h), y = (int)(j - (size_t)k * h);
    const int cols[4] = {0, min(1, w - 1), max(w - 2, 0), w - 1};
    c = labels[(size_t)y * w + cols[k]];
  }
  pairs[i] = c ? make_uint2(c, uf_find(parent, c)) : make_uint2(0u, 0u);
}

hipError_t block_colour_roots2d(hipStream_t s, const uint32_t *labels, int h, int w, uint32_t *parent, uint2 *pairs, size_t n_pairs) {
  if (n_pairs == 0) return hipSuccess;
  k_block_colour_roots2d<<<(unsigned)((n_pairs + 255) / 256), 256, 0, s>>>(labels, h, w, parent, pairs, n_pairs);
  return hipGetLastError();
}

// pairs[i] = (colour, root of colour) for the pixels of local rows 0, 1, h - 2, h - 1 (4 w pairs; (0, 0) where uncoloured)
__global__ void k_block_colour_roots(const uint32_t *__restrict__ labels, int h, int w, uint32_t *parent, uint2 *pairs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)4 * w) return;
  const int k = (int)(i / (size_t)w), x = (int)(i - (size_t)k * w);
  const int rows[4] = {0, min(1, h - 1), max(h - 2, 0), h - 1};
  const uint32_t c = labels[(size_t)rows[k] * w + x];
  pairs[i] = c ? make_uint2(c, uf_find(parent, c)) : make_uint2(0u, 0u);
}

hipError_t block_colour_roots(hipStream_t s, const uint32_t *labels, int h, int w, uint32_t *parent, uint2 *pairs) {
  if (h == 0 || w == 0) return hipSuccess;
  k_block_colour_roots<<<(unsigned)((4 * (size_t)w + 255) / 256), 256, 0, s>>>(labels, h, w, parent, pairs);
  return hipGetLastError();
}

// ---- final-only path ---------------------------------------------------------------------

// One launch, 64x64 tiles.  Joining every crossing pixel pair in the global forest costs ~60 M
// unions that all end at the one surviving root; instead each tile first joins its own coloured
// pixels in an LDS union-find (adjacency rule of find_merge: one endpoint of a pair must be an
// interior pixel, lib.rs:411-434), takes the smallest colour of every local component, and only then
// touches the global forest: once per colour REGION of a component (its top-left pixels) and once
// per crossing pair on the tile's right / bottom edge -- ~6x fewer global unions, most of them local.
constexpr int UT = 64;                                   // tile side
constexpr int UT_PX = UT * UT / 256;                     // pixels per thread (16): column strips as in k_resolve_local

__device__ __forceinline__ uint32_t lds_find(uint32_t *P, uint32_t x) {
  for (;;) {
    const uint32_t p = P[x];
    if (p == x) return x;
    const uint32_t g = P[p];
    if (g == p) return p;
    atomicMin(&P[x], g);
    x = g;
  }
}
__device__ __forceinline__ void lds_union(uint32_t *P, uint32_t a, uint32_t b) {
  for (;;) {
    a = lds_find(P, a);
    b = lds_find(P, b);
    if (a == b) return;
    if (a > b) { const uint32_t t = a; a = b; b = t; }
    const uint32_t old = atomicCAS(&P[b], b, a);
    if (old == b) return;
    b = old;
  }
}

__global__ __launch_bounds__(256) void k_union_tiles(const uint32_t *__restrict__ labels, int H, int W, int tilesX, uint32_t *parent,
                                                     const uint32_t *__restrict__ tile_min) {
  if (tile_min[blockIdx.x] != 0u) return;          // a one-lake tile: k_tile_scan and k_union_seeds deal with it
  __shared__ uint32_t sP[UT * UT];        // local forest over the tile's pixels
  __shared__ uint32_t sMin[UT * UT];      // smallest colour of a local component, kept at its root
  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const int tid = threadIdx.x, lane = tid & 63, strip = tid >> 6;
  const int x0 = tile_x * UT, y0 = tile_y * UT;
  const int gx = x0 + lane, gy0 = y0 + strip * UT_PX;
  const int gxc = min(gx, W - 1);

  uint32_t col[UT_PX], colR[UT_PX];       // colour of the pixel / of its right neighbour; 0 = uncoloured or outside
  uint32_t colD_last;                     // colour below the strip's last pixel
#pragma unroll
  for (int i = 0; i < UT_PX; ++i) {
    const int gy = gy0 + i, gyc = min(gy, H - 1);
    const size_t g = (size_t)gyc * W + gxc;
    // final level: a pixel is coloured exactly when its label is non-zero (the stamps are not read)
    const uint32_t l = labels[g];
    const uint32_t lr = labels[(size_t)gyc * W + min(gx + 1, W - 1)];
    col[i] = (gy < H && gx < W) ? l : 0u;
    colR[i] = (gy < H && gx + 1 < W) ? lr : 0u;
    sP[(strip * UT_PX + i) * UT + lane] = (uint32_t)((strip * UT_PX + i) * UT + lane);
    sMin[(strip * UT_PX + i) * UT + lane] = 0xFFFFFFFFu;
  }
  {
    const int gy = gy0 + UT_PX, gyc = min(gy, H - 1);
    const uint32_t l = labels[(size_t)gyc * W + gxc];
    colD_last = (gy < H && gx < W) ? l : 0u;
  }
  __syncthreads();

  // local unions: right and down neighbours inside the tile (any colours: touching lakes merge)
#pragma unroll
  for (int i = 0; i < UT_PX; ++i) {
    if (col[i] == 0u) continue;
    const int ly = strip * UT_PX + i, gy = gy0 + i;
    const bool ip = interior(gy, gx, H, W);
    const uint32_t cell = (uint32_t)(ly * UT + lane);
    if (lane + 1 < UT && colR[i] != 0u && (ip || interior(gy, gx + 1, H, W))) lds_union(sP, cell, cell + 1);
    const uint32_t cd = i == UT_PX - 1 ? colD_last : col[i + 1];
    if (ly + 1 < UT && cd != 0u && (ip || interior(gy + 1, gx, H, W))) lds_union(sP, cell, cell + UT);
  }
  __syncthreads();
  uint32_t root[UT_PX];
#pragma unroll
  for (int i = 0; i < UT_PX; ++i) {
    root[i] = col[i] ? lds_find(sP, (uint32_t)((strip * UT_PX + i) * UT + lane)) : 0u;
    if (col[i]) atomicMin(&sMin[root[i]], col[i]);
  }
  __syncthreads();

  // colour of the left neighbour, fetched with every lane active (a shuffle under a divergent
  // branch would read inactive lanes)
  uint32_t colL[UT_PX];
#pragma unroll
  for (int i = 0; i < UT_PX; ++i) {
    const uint32_t v = __shfl_up(col[i], 1, 64);
    colL[i] = lane > 0 ? v : 0u;
  }

  // global unions
#pragma unroll
  for (int i = 0; i < UT_PX; ++i) {
    if (col[i] == 0u) continue;
    const int ly = strip * UT_PX + i, gy = gy0 + i;
    const uint32_t cmin = sMin[root[i]];
    // (a) one union per colour region of the component: the pixel has no same-coloured pixel of its
    //     own tile to the left, nor above inside this thread's strip (a region's top-left pixel
    //     always qualifies; the strip above belongs to another wave and counts as "different")
    const uint32_t cu = i > 0 ? col[i - 1] : 0u;
    if (col[i] != cmin && colL[i] != col[i] && cu != col[i]) uf_union(parent, col[i], cmin);
    // (b) crossing pairs over the tile's right and bottom edge
    const bool ip = interior(gy, gx, H, W);
    if (lane == UT - 1 && colR[i] != 0u && colR[i] != col[i] && (ip || interior(gy, gx + 1, H, W))) uf_union(parent, col[i], colR[i]);
    if (ly == UT - 1 && colD_last != 0u && colD_last != col[i] && (ip || interior(gy + 1, gx, H, W))) uf_union(parent, col[i], colD_last);
  }
}

typedef uint32_t u32x4_m __attribute__((ext_vector_type(4)));

// the four corner pixels of the image touch border pixels only: a seed there never joins anything
__device__ __forceinline__ bool image_corner(int y, int x, int H, int W) { return (y == 0 || y == H - 1) && (x == 0 || x == W - 1); }

// ---- one-lake tiles -------------------------------------------------------------------------
//
// At the final level of an ordinary field nearly every 64 x 64 tile is a single lake: every pixel of it
// that is an interior pixel of the image is coloured.  (Those pixels form a rectangle, so they are
// 4-connected; a coloured pixel of the image border joins through its inward neighbour, which the
// adjacency rule of find_merge allows: one endpoint interior, lib.rs:411-434.)  For such a tile nothing
// has to be discovered locally, and two such tiles that share an edge are always joined across it.
// What remains is bookkeeping, split so that no thread idles behind another's union latency and no
// crowd of unions walks the same chain at once:
//   k_tile_scan    per tile: "one lake?" and its smallest colour cmin -> tile_min (0 = no)
//   k_tile_links   per tile: joins cmin with the cmin of the one-lake tile to the right / below;
//                  k_tile_roots then replaces every tile's cmin by the root of its lake
//   k_union_seeds  per SEED: colour i+1 joins the cmin of the tile its seed pixel lies in
//   k_tile_edges   per tile edge pixel: a colour a tile holds WITHOUT its seed came in over an edge; if from a
//                  one-lake tile, the two tiles are joined and that tile holds the colour already (induction
//                  along the colour's region); if from a general tile, the colour joins cmin here, as do the
//                  colours across the right / bottom edge when the tile there is general
//   k_union_tiles  the general path (LDS union-find) for the tiles k_tile_scan turned down
// preclassified: the resolve kernel has filled tile_min already; only its "undecided" tiles are looked at
__global__ __launch_bounds__(256) void k_tile_scan(const uint32_t *__restrict__ labels, int H, int W, int tilesX, uint32_t *tile_min,
                                                   int preclassified) {
  __shared__ uint32_t sWaveMin[4];
  if (preclassified && tile_min[blockIdx.x] != 0xFFFFFFFFu) return;
  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int x0 = tile_x * UT, y0 = tile_y * UT;
  // 16 pixels per thread: patch (tid & 15, tid >> 4); one 16-byte load per row when the rows are aligned and
  // the patch lies inside the plane
  const int gx0 = x0 + (tid & 15) * 4, gy0 = y0 + (tid >> 4) * 4;
  const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(labels) & 15u) == 0 && gx0 + 4 <= W;
  uint32_t tmin1 = 0xFFFFFFFFu;      // smallest (colour - 1): an uncoloured pixel (0) wraps to the top and never wins
  bool ok = true;                    // every in-plane image-interior pixel seen is coloured
  bool any_interior = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int gy = gy0 + r, gyc = min(gy, H - 1);
    uint32_t v[4];
    if (vec) {
      const u32x4_m q = *reinterpret_cast<const u32x4_m *>(labels + (size_t)gyc * W + gx0);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = labels[(size_t)gyc * W + min(gx0 + c, W - 1)];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int gx = gx0 + c;
      const bool in_plane = gy < H && gx < W;
      const bool inter = in_plane && interior(gy, gx, H, W);
      if (in_plane && !image_corner(gy, gx, H, W)) tmin1 = min(tmin1, v[c] - 1u);
      ok = ok && (!inter || v[c] != 0u);
      any_interior = any_interior || inter;
    }
  }
  const bool one_lake = __syncthreads_and(ok) != 0 && __syncthreads_or(any_interior) != 0;
  for (int o = 32; o > 0; o >>= 1) tmin1 = min(tmin1, (uint32_t)__shfl_xor(tmin1, o, 64));
  if (lane == 0) sWaveMin[wave] = tmin1;
  __syncthreads();
  if (tid == 0) tile_min[blockIdx.x] = one_lake ? min(min(sWaveMin[0], sWaveMin[1]), min(sWaveMin[2], sWaveMin[3])) + 1u : 0u;
}

// Every pair of one-lake tiles that share an edge must end in one set.  One union per pair (32 k of them at 8192^2, nearly
// all into the same set) was 43 us of CAS retries on a few hot roots.  Instead: along a row, a RUN of one-lake tiles is
// tied together by a segmented min-scan in the wave -- tile i joins the smallest colour of the run's tiles before it, a
// child slot of its own and a chain a few hooks deep (the prefix minima of the run) -- and two runs in consecutive rows
// are joined once, at the leftmost column they share (the head of one of them).
__global__ void k_tile_links(const uint32_t *__restrict__ tile_min, int tilesX, int tilesY, uint32_t *parent) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool in = t < tilesX * tilesY;
  const int tx = in ? t % tilesX : 0, ty = in ? t / tilesX : 0;
  const uint32_t m = in ? tile_min[t] : 0u;
  const uint32_t left = in && tx > 0 ? tile_min[t - 1] : 0u;
  const uint32_t md = in && ty + 1 < tilesY ? tile_min[t + tilesX] : 0u;
  const uint32_t md_left = in && ty + 1 < tilesY && tx > 0 ? tile_min[t + tilesX - 1] : 0u;
  const bool head = m != 0u && left == 0u, md_head = md != 0u && md_left == 0u;
  // inclusive min over the run, from its first tile IN THIS WAVE to this lane (a tile that is not one lake is a segment
  // of its own: its value is never used)
  uint32_t v = m != 0u ? m : 0xFFFFFFFFu;
  bool f = head || m == 0u || lane == 0;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t pv = (uint32_t)__shfl_up((int)v, d, 64);
    const int pf = __shfl_up((int)f, d, 64);
    if (lane >= d && !f) { v = min(v, pv); f = pf != 0; }
  }
  const uint32_t before = (uint32_t)__shfl_up((int)v, 1, 64);      // the run's smallest colour up to the tile on the left
  if (m == 0u) return;
  if (!head) {
    const uint32_t join = lane == 0 ? left : before;               // (lane 0: the run goes on from the wave before)
    if (join != m) uf_union(parent, m, join);
  }
  if (md != 0u && md != m && (head || md_head)) uf_union(parent, m, md);
}

// after the links: every one-lake tile remembers the ROOT of its lake instead of its own smallest colour, so
// that the 450 seeds of a tile find a root in one load instead of all walking (and compressing) the same chain
// (a plain walk: the forest is at rest between two launches, and 16 k halving finds on the links' few chains were 29 us
// of atomics on the same words against 6 us for the walk)
// mark (optional, zeroed, one word per colour): mark[c] = 1 for every colour the tile links have touched -- a tile's
// smallest colour or a lake's root; every other colour is still its own root and nobody's parent
__global__ void k_tile_roots(uint32_t *tile_min, int ntiles, uint32_t *parent, uint32_t *mark) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ntiles) return;
  const uint32_t m = tile_min[t];
  if (m != 0u && m != 0xFFFFFFFFu) {
    const uint32_t r = uf_root(parent, m);
    tile_min[t] = r;
    if (mark) { mark[r] = 1u; mark[m] = 1u; }
  }
}

// Nearly every seed is a colour nobody has touched yet (its own root, no child: k_tile_roots marks the colours the tile
// links have touched) that goes under its tile's root, a smaller colour: for those the hook is a PLAIN store, without
// reading the slot -- nobody else reads or writes it in this launch: a union names a colour only as the seed's own (this
// thread) or as a tile root (marked), and a find only passes through colours that something was hooked under.  Every
// other case takes the union.  (54 -> 45 us at 8192^2: what is left is the 16 B per seed the kernel moves.)
__global__ void k_union_seeds(const uint32_t *__restrict__ seeds_rc, size_t n_seeds, int H, int W, int tilesX,
                              const uint32_t *__restrict__ tile_min, uint32_t *parent, const uint32_t *__restrict__ tile_root_mark) {
  // (four seeds per thread and round, each stage's loads in flight together: the kernel is three dependent loads and a
  // store per seed, and with one seed per thread it was 54 us of waiting for them)
  constexpr int SU = 4;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n_seeds; i0 += SU * step) {
    uint2 rc[SU];
    uint32_t cmin[SU], mark[SU];
    bool ok[SU];
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const size_t i = i0 + u * step;
      rc[u] = reinterpret_cast<const uint2 *>(seeds_rc)[i < n_seeds ? i : n_seeds - 1];
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const size_t i = i0 + u * step;
      // (a seed outside the plane: the transform has reported it; here it only must not index outside tile_min)
      ok[u] = i < n_seeds && rc[u].x < (uint32_t)H && rc[u].y < (uint32_t)W;
      const size_t me = ok[u] ? i + 1 : 0;
      cmin[u] = tile_min[ok[u] ? (size_t)(rc[u].x / UT) * tilesX + rc[u].y / UT : 0];
      mark[u] = tile_root_mark ? tile_root_mark[me] : 1u;
    }
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const uint32_t me = (uint32_t)(i0 + u * step + 1);
      if (!ok[u] || cmin[u] == 0u || cmin[u] == me || image_corner((int)rc[u].x, (int)rc[u].y, H, W)) continue;
      if (cmin[u] < me && mark[u] == 0u) parent[me] = cmin[u];      // (unmarked: parent[me] == me still, see k_tile_roots)
      else uf_union(parent, me, cmin[u]);
    }
  }
}

// wave = edge of the tile (0 bottom, 1 right, 2 top, 3 left), lane = position along it
__global__ __launch_bounds__(256) void k_tile_edges(const uint32_t *__restrict__ labels, int H, int W, int tilesX, int tilesY,
                                                    const uint32_t *__restrict__ tile_min, uint32_t *parent) {
  const uint32_t cmin = tile_min[blockIdx.x];
  if (cmin == 0u) return;
  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the tile across this wave's edge (none at the image border).  Only a GENERAL tile there gives this wave
  // anything to do -- checked before a single pixel is loaded: a tile edge is one pixel per row, a cache line
  // fetched for each (the kernel moved 400 MB to look at 16 MB of pixels and find, on an ordinary field, nothing).
  const int nb = wave == 0 ? (tile_y + 1 < tilesY ? (int)blockIdx.x + tilesX : -1)
               : wave == 1 ? (tile_x + 1 < tilesX ? (int)blockIdx.x + 1 : -1)
               : wave == 2 ? (tile_y > 0 ? (int)blockIdx.x - tilesX : -1)
                           : (tile_x > 0 ? (int)blockIdx.x - 1 : -1);
  if (nb < 0 || tile_min[nb] != 0u) return;
  const int x0 = tile_x * UT, y0 = tile_y * UT;
  const int ylast = min(y0 + UT, H) - 1, xlast = min(x0 + UT, W) - 1;       // the tile's last row / column inside the plane
  const int ey = wave == 0 ? ylast : (wave == 2 ? y0 : y0 + lane);
  const int ex = wave == 1 ? xlast : (wave == 3 ? x0 : x0 + lane);
  const bool own_ok = ey <= ylast && ex <= xlast;
  const int fy = ey + (wave == 0 ? 1 : 0), fx = ex + (wave == 1 ? 1 : 0);
  const bool far_ok = own_ok && wave < 2 && fy < H && fx < W;
  // unconditional loads on clamped addresses (ws_relax.hip)
  const uint32_t own_v = labels[(size_t)min(ey, H - 1) * W + min(ex, W - 1)];
  const uint32_t far_v = labels[(size_t)min(fy, H - 1) * W + min(fx, W - 1)];
  const uint32_t own = own_ok && !image_corner(ey, ex, H, W) ? own_v : 0u, far = far_ok ? far_v : 0u;
  // own colours on this edge join cmin: a colour whose seed lies elsewhere reached this tile over one of its
  // edges, from a one-lake tile (joined with this one by k_tile_links, and holding the colour already, by
  // induction along the colour's region) or from a general one -- this case.  One union per run of equal colours.
  const uint32_t want_own = (own != 0u && own != cmin) ? own : 0u;
  const uint32_t want_own_prev = __shfl_up(want_own, 1, 64);
  if (want_own != 0u && (lane == 0 || want_own != want_own_prev)) uf_union(parent, want_own, cmin);
  // colours across the right / bottom edge (the general tile's own kernel only looks right and down)
  const bool pair_ok = own != 0u && far != 0u && (interior(ey, ex, H, W) || interior(fy, fx, H, W));
  const uint32_t want_far = (wave < 2 && pair_ok && far != cmin) ? far : 0u;
  const uint32_t want_far_prev = __shfl_up(want_far, 1, 64);
  if (want_far != 0u && (lane == 0 || want_far != want_far_prev)) uf_union(parent, want_far, cmin);
}

// tile_min: union_image_tiles(h, w) words of scratch
hipError_t union_image(hipStream_t s, const uint32_t *labels, const uint32_t *seeds_rc, size_t n_seeds, int h, int w,
                       uint32_t *parent, uint32_t *tile_min, bool preclassified, uint32_t *tile_root_mark) {
  if (h == 0 || w == 0) return hipSuccess;
  const int tx = (w + UT - 1) / UT, ty = (h + UT - 1) / UT;
  hipError_t e;
  k_tile_scan<<<tx * ty, 256, 0, s>>>(labels, h, w, tx, tile_min, preclassified ? 1 : 0);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  k_tile_links<<<(tx * ty + 255) / 256, 256, 0, s>>>(tile_min, tx, ty, parent);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  k_tile_roots<<<(tx * ty + 255) / 256, 256, 0, s>>>(tile_min, tx * ty, parent, tile_root_mark);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (n_seeds) {
    const int blocks = (int)std::min<size_t>((n_seeds + 1023) / 1024, 16384);      // four seeds per thread
    k_union_seeds<<<blocks, 256, 0, s>>>(seeds_rc, n_seeds, h, w, tx, tile_min, parent, tile_root_mark);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    k_tile_edges<<<tx * ty, 256, 0, s>>>(labels, h, w, tx, ty, tile_min, parent);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  k_union_tiles<<<tx * ty, 256, 0, s>>>(labels, h, w, tx, parent, tile_min);
  return hipGetLastError();
}
size_t union_image_tiles(int h, int w) { return (size_t)((w + UT - 1) / UT) * ((h + UT - 1) / UT); }

// After the last union: every colour points straight at its root, so that relabelling is one gather.
__global__ void k_uf_flatten(uint32_t *parent, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t r = uf_find(parent, (uint32_t)i);
    if (r != (uint32_t)i) atomicMin(parent + i, r);
  }
}

// lib.rs:589-592 with the closed, flattened map, final level: uncoloured pixels carry label 0 and parent[0] == 0
__global__ void k_relabel_flat(const uint32_t *__restrict__ labels, const uint32_t *__restrict__ parent, uint32_t *out, size_t n) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  const bool vec = ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  const size_t nv = vec ? n / 4 : 0;
  for (size_t i = tid; i < nv; i += step) {
    const u32x4_m l = reinterpret_cast<const u32x4_m *>(labels)[i];
    reinterpret_cast<u32x4_m *>(out)[i] = u32x4_m{parent[l.x], parent[l.y], parent[l.z], parent[l.w]};
  }
  for (size_t i = nv * 4 + tid; i < n; i += step) out[i] = parent[labels[i]];
}

// The same tile by tile, for a plane whose 64 x 64 tiles union_image has classified: a one-lake tile strictly inside the
// image (every pixel coloured, all of one lake: tile_min = a colour of it) is FILLED with its root -- no label is read;
// every other tile takes the gather (of roots: uf_root).  At the final level of a map that floods completely nearly every tile is one lake:
// 8192^2 bench field 121 -> ~60 us (the gather reads and writes the plane, 537 MB; the fill writes 268 MB).
__global__ __launch_bounds__(256) void k_relabel_tiles(const uint32_t *__restrict__ labels, const uint32_t *__restrict__ parent,
                                                       const uint32_t *__restrict__ tile_min, uint32_t *out, int H, int W, int tilesX) {
  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const int x0 = tile_x * UT, y0 = tile_y * UT;
  const int gx0 = x0 + (threadIdx.x & 15) * 4, gy0 = y0 + (threadIdx.x >> 4) * 4;
  const uint32_t cm = tile_min[blockIdx.x];
  const bool inside = x0 >= 1 && x0 + UT <= W - 1 && y0 >= 1 && y0 + UT <= H - 1;      // workgroup uniform
  const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(labels) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
  if (cm != 0u && cm != 0xFFFFFFFFu && inside && vec) {
    const uint32_t r = cm;      // (a root: relabel_final_u32 runs k_tile_roots first)
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<u32x4_m *>(out + (size_t)(gy0 + k) * W + gx0) = u32x4_m{r, r, r, r};
    return;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int gy = gy0 + k;
    if (gy >= H) break;
    if (vec && gx0 + 4 <= W) {
      const u32x4_m l = *reinterpret_cast<const u32x4_m *>(labels + (size_t)gy * W + gx0);
      *reinterpret_cast<u32x4_m *>(out + (size_t)gy * W + gx0) = u32x4_m{uf_root(parent, l.x), uf_root(parent, l.y), uf_root(parent, l.z), uf_root(parent, l.w)};
    } else {
      for (int c = 0; c < 4; ++c)
        if (gx0 + c < W) out[(size_t)gy * W + gx0 + c] = uf_root(parent, labels[(size_t)gy * W + gx0 + c]);
    }
  }
}

hipError_t relabel_final_u32(hipStream_t s, const uint32_t *labels, uint32_t *parent, size_t n_colours, uint32_t *out, size_t n,
                             uint32_t *tile_min, int h, int w) {
  if (n == 0) return hipSuccess;
  if (tile_min && (size_t)h * w == n) {
    // (no flattening pass over the 7 M colours first -- k_uf_flatten was 21 us at 8192^2: a one-lake tile needs ONE root,
    // found once per tile by k_tile_roots -- the seeds' unions hook the lake's root under ever smaller colours, a chain
    // ~ln(n) hooks deep that every tile would otherwise walk -- and the other tiles walk what path halving left)
    const int tx = (w + UT - 1) / UT, ty = (h + UT - 1) / UT;
    k_tile_roots<<<(tx * ty + 255) / 256, 256, 0, s>>>(tile_min, tx * ty, parent, nullptr);
    hipError_t e0 = hipGetLastError();
    if (e0 != hipSuccess) return e0;
    k_relabel_tiles<<<tx * ty, 256, 0, s>>>(labels, parent, tile_min, out, h, w, tx);
    return hipGetLastError();
  }
  const int fb = (int)std::min<size_t>((n_colours + 255) / 256, 8192);
  k_uf_flatten<<<fb > 0 ? fb : 1, 256, 0, s>>>(parent, n_colours);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int blocks = (int)std::min<size_t>((n / 4 + 255) / 256 + 1, 16384);
  k_relabel_flat<<<blocks, 256, 0, s>>>(labels, parent, out, n);
  return hipGetLastError();
}

template <typename OutT>
__global__ void k_relabel(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels, uint32_t *parent,
                          OutT *out, size_t n, uint32_t level) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t k = keys[i];
    OutT v = 0;
    if (k != KEY_INF && (k >> 24) <= level) v = (OutT)uf_find(parent, labels[i]);    // lib.rs:589-592 with the closed map
    out[i] = v;
  }
}

hipError_t relabel_u32(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *parent,
                       uint32_t *out, size_t n, uint32_t level) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_relabel<uint32_t><<<blocks, 256, 0, s>>>(keys, labels, parent, out, n, level);
  return hipGetLastError();
}

hipError_t relabel_u64(hipStream_t s, const uint32_t *keys, const uint32_t *labels, uint32_t *parent,
                       uint64_t *out, size_t n, uint32_t level) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 1023) / 1024 < 8192 ? (n + 1023) / 1024 : 8192);
  k_relabel<uint64_t><<<blocks, 256, 0, s>>>(keys, labels, parent, out, n, level);
  return hipGetLastError();
}

}  // namespace wsk
