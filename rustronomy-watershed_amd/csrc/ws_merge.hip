// ws_merge.hip -- gfx950 kernels of the merging transform (see ws_merge.hpp for the maths).
//
// Union-find over seed colours, lock-free, union-by-min-index: a root only ever gets hooked
// under a SMALLER index (atomicCAS on the root's own slot) and path halving only ever
// lowers a parent (atomicMin), so every parent chain is monotone and a stale read is still
// an ancestor.  All parent updates are device-scope atomics: XCD L2s are not coherent with
// each other, and nothing here relies on a plain store being seen inside a launch.
#include "ws_common.hpp"
#include "ws_merge.hpp"

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

__device__ __forceinline__ void gather_items(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                             int H, int W, int y, int x0, PixelItems &it) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    it.px_lvl[k] = it.er_lvl[k] = it.ed_lvl[k] = 0xFFFFFFFFu;
    const int x = x0 + k;
    if (x >= W) continue;
    const size_t p = (size_t)y * W + x;
    const uint32_t kp = keys[p];
    if (kp == KEY_INF) continue;                       // uncoloured pixels carry no lake
    const uint32_t lp = labels[p], vp = kp >> 24;
    it.px_lvl[k] = vp;
    it.px_col[k] = lp;
    const bool ip = interior(y, x, H, W);
    if (x + 1 < W) {
      const uint32_t kq = keys[p + 1];
      if (kq != KEY_INF) {
        const uint32_t lq = labels[p + 1];
        // find_merge only sees pairs around a 3x3 window centre (lib.rs:411-434)
        if (lq != lp && (ip || interior(y, x + 1, H, W))) {
          it.er_lvl[k] = max(vp, kq >> 24);
          it.er[k] = make_uint2(lp, lq);
        }
      }
    }
    if (y + 1 < H) {
      const uint32_t kq = keys[p + W];
      if (kq != KEY_INF) {
        const uint32_t lq = labels[p + W];
        if (lq != lp && (ip || interior(y + 1, x, H, W))) {
          it.ed_lvl[k] = max(vp, kq >> 24);
          it.ed[k] = make_uint2(lp, lq);
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_level_hist(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                                    int H, int W, int segs, u64c *hist_px, u64c *hist_edge) {
  __shared__ uint32_t s_px[NLEVELS], s_ed[NLEVELS];
  s_px[threadIdx.x] = 0;
  s_ed[threadIdx.x] = 0;
  __syncthreads();
  const int y = blockIdx.x / segs, seg = blockIdx.x % segs;
  PixelItems it;
  gather_items(keys, labels, H, W, y, seg * MSEG + threadIdx.x * 4, it);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (it.px_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_px[it.px_lvl[k]], 1u);
    if (it.er_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.er_lvl[k]], 1u);
    if (it.ed_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.ed_lvl[k]], 1u);
  }
  __syncthreads();
  if (s_px[threadIdx.x]) atomicAdd(&hist_px[threadIdx.x], (u64c)s_px[threadIdx.x]);
  if (s_ed[threadIdx.x]) atomicAdd(&hist_edge[threadIdx.x], (u64c)s_ed[threadIdx.x]);
}

hipError_t level_hist(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                      u64c *hist_px, u64c *hist_edge) {
  if (h == 0 || w == 0) return hipSuccess;
  const int segs = (w + MSEG - 1) / MSEG;
  k_level_hist<<<h * segs, 256, 0, s>>>(keys, labels, h, w, segs, hist_px, hist_edge);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_level_scatter(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                                       int H, int W, int segs, u64c *cursor_px, u64c *cursor_edge,
                                                       uint32_t *px_items, uint2 *edge_items) {
  __shared__ uint32_t s_px[NLEVELS], s_ed[NLEVELS];
  __shared__ u64c s_bpx[NLEVELS], s_bed[NLEVELS];
  s_px[threadIdx.x] = 0;
  s_ed[threadIdx.x] = 0;
  __syncthreads();
  const int y = blockIdx.x / segs, seg = blockIdx.x % segs;
  PixelItems it;
  gather_items(keys, labels, H, W, y, seg * MSEG + threadIdx.x * 4, it);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (it.px_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_px[it.px_lvl[k]], 1u);
    if (it.er_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.er_lvl[k]], 1u);
    if (it.ed_lvl[k] != 0xFFFFFFFFu) atomicAdd(&s_ed[it.ed_lvl[k]], 1u);
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

hipError_t level_scatter(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                         u64c *cursor_px, u64c *cursor_edge, uint32_t *px_items, uint2 *edge_items) {
  if (h == 0 || w == 0) return hipSuccess;
  const int segs = (w + MSEG - 1) / MSEG;
  k_level_scatter<<<h * segs, 256, 0, s>>>(keys, labels, h, w, segs, cursor_px, cursor_edge, px_items, edge_items);
  return hipGetLastError();
}

// ---- per-level union / sizes / emit -----------------------------------------------------

__global__ void k_union_edges(const uint2 *__restrict__ edges, size_t n, uint32_t *parent, uint32_t *hooked,
                              uint32_t *hooked_count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint2 e = edges[i];
    const uint32_t lost = uf_union(parent, e.x, e.y);
    if (lost != 0xFFFFFFFFu && hooked) hooked[atomicAdd(hooked_count, 1u)] = lost;
  }
}

hipError_t union_edges(hipStream_t s, const uint2 *edges, size_t n, uint32_t *parent, uint32_t *hooked,
                       uint32_t *hooked_count) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  k_union_edges<<<blocks, 256, 0, s>>>(edges, n, parent, hooked, hooked_count);
  return hipGetLastError();
}

// after a level's unions: every node hooked in this level hands its accumulated area to its final root
__global__ void k_fold_sizes(const uint32_t *__restrict__ hooked, const uint32_t *__restrict__ hooked_count,
                             uint32_t *parent, uint32_t *size) {
  const uint32_t n = *hooked_count;
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t step = gridDim.x * blockDim.x;
  for (; i < n; i += step) {
    const uint32_t b = hooked[i];
    const uint32_t area = size[b];
    if (area) atomicAdd(&size[uf_find(parent, b)], area);
  }
}

hipError_t fold_sizes(hipStream_t s, const uint32_t *hooked, const uint32_t *hooked_count, uint32_t *parent,
                      uint32_t *size) {
  k_fold_sizes<<<512, 256, 0, s>>>(hooked, hooked_count, parent, size);
  return hipGetLastError();
}

__global__ void k_add_arrivals(const uint32_t *__restrict__ px_items, size_t n, uint32_t *parent, uint32_t *size) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += step) atomicAdd(&size[uf_find(parent, px_items[i])], 1u);
}

hipError_t add_arrivals(hipStream_t s, const uint32_t *px_items, size_t n, uint32_t *parent, uint32_t *size) {
  if (n == 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
  k_add_arrivals<<<blocks, 256, 0, s>>>(px_items, n, parent, size);
  return hipGetLastError();
}

// lib.rs:628-635 sparsely: one (colour, area) record per lake with area > 0
__global__ void k_emit_lakes(const uint32_t *__restrict__ parent, const uint32_t *__restrict__ size, size_t n_colours,
                             uint64_t *lakes, size_t cap, u64c *cursor) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + 1;     // colour 0 = uncoloured
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n_colours; i += step) {
    const uint32_t area = size[i];
    if (parent[i] == (uint32_t)i && area) {
      const u64c pos = atomicAdd(cursor, 1ull);
      if (pos < cap) { lakes[2 * pos] = (uint64_t)i; lakes[2 * pos + 1] = (uint64_t)area; }
    }
  }
}

hipError_t emit_lakes(hipStream_t s, const uint32_t *parent, const uint32_t *size, size_t n_colours,
                      uint64_t *lakes, size_t cap, u64c *cursor) {
  if (n_colours <= 1) return hipSuccess;
  const int blocks = (int)((n_colours + 255) / 256 < 4096 ? (n_colours + 255) / 256 : 4096);
  k_emit_lakes<<<blocks, 256, 0, s>>>(parent, size, n_colours, lakes, cap, cursor);
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

__global__ __launch_bounds__(256) void k_union_tiles(const uint32_t *__restrict__ keys, const uint32_t *__restrict__ labels,
                                                     int H, int W, int tilesX, uint32_t *parent) {
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
    const uint32_t k = keys[g], l = labels[g];
    const uint32_t kr = keys[(size_t)gyc * W + min(gx + 1, W - 1)], lr = labels[(size_t)gyc * W + min(gx + 1, W - 1)];
    col[i] = (gy < H && gx < W && k != KEY_INF) ? l : 0u;
    colR[i] = (gy < H && gx + 1 < W && kr != KEY_INF) ? lr : 0u;
    sP[(strip * UT_PX + i) * UT + lane] = (uint32_t)((strip * UT_PX + i) * UT + lane);
    sMin[(strip * UT_PX + i) * UT + lane] = 0xFFFFFFFFu;
  }
  {
    const int gy = gy0 + UT_PX, gyc = min(gy, H - 1);
    const uint32_t k = keys[(size_t)gyc * W + gxc], l = labels[(size_t)gyc * W + gxc];
    colD_last = (gy < H && gx < W && k != KEY_INF) ? l : 0u;
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

hipError_t union_image(hipStream_t s, const uint32_t *keys, const uint32_t *labels, int h, int w,
                       uint32_t *parent) {
  if (h == 0 || w == 0) return hipSuccess;
  const int tx = (w + UT - 1) / UT, ty = (h + UT - 1) / UT;
  k_union_tiles<<<tx * ty, 256, 0, s>>>(keys, labels, h, w, tx, parent);
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
