// ws_block.hip -- one field tiled over several GPUs in row blocks: the kernels around the label hand-over.
//
// A rank's plane is its own rows plus one halo row per neighbour.  Stamps need an iteration (relax locally, swap halo
// rows, repeat while a halo changed: distributed.py); labels do not: after the local resolve every label of a block is a
// colour or a reference to a halo pixel, i.e. to a pixel of a neighbour's BOUNDARY row (its first or last own row).  So
// the boundary rows of all ranks form a closed table -- entry (rank r, side s, column x) at (2 r + s) w + x, s = 0 for
// the first own row, 1 for the last -- in which every entry is a colour or a reference to another entry.  One
// all-gather of 2 w words per rank, and every rank resolves the table for itself: no rounds, no host decisions.
#include "ws_common.hpp"

#include <algorithm>

namespace wsk {

namespace {
constexpr uint32_t REF = 0x80000000u;      // as in ws_kernels.hip: REF | pixel index

__global__ void k_block_export(const uint32_t *__restrict__ labels, int h, int w, int halo_flags, uint32_t rank, uint32_t *rows) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)2 * w) return;
  const int side = i >= (size_t)w, col = (int)(i - (size_t)side * w);
  const int row = side == 0 ? ((halo_flags & 1) ? 1 : 0) : ((halo_flags & 2) ? h - 2 : h - 1);
  uint32_t v = labels[(size_t)row * w + col];
  if (v & REF) {      // after the chase a reference can only name a halo pixel: (row 0 | h - 1, column c)
    const size_t idx = v & ~REF;
    const int r = (int)(idx / (size_t)w), c = (int)(idx - (size_t)r * w);
    if (r == 0 && (halo_flags & 1)) v = REF | (uint32_t)(((size_t)(rank - 1) * 2 + 1) * w + c);            // the upper neighbour's last own row
    else if (r == h - 1 && (halo_flags & 2)) v = REF | (uint32_t)(((size_t)(rank + 1) * 2 + 0) * w + c);   // the lower neighbour's first own row
    else v = 0u;      // not reachable: k_resolve_chase leaves references to halo pixels only
  }
  rows[i] = v;
}

// A reference names an entry with a strictly smaller arrival stamp, so chains end (at a seed's colour, or at 0 for a
// pixel no flood reached); the input table is never written, every thread walks its own chain.
__global__ void k_table_jump(const uint32_t *__restrict__ table, size_t n, uint32_t *resolved) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t v = table[i];
  for (size_t hops = 0; (v & REF) && (size_t)(v & ~REF) < n && hops < n; ++hops) v = table[v & ~REF];
  resolved[i] = (v & REF) ? 0u : v;
}

__global__ void k_fill_halo_rows(const uint32_t *__restrict__ resolved, uint32_t rank, uint32_t *labels, int h, int w, int halo_flags) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= w) return;
  if (halo_flags & 1) labels[x] = resolved[((size_t)(rank - 1) * 2 + 1) * w + x];
  if (halo_flags & 2) labels[(size_t)(h - 1) * w + x] = resolved[((size_t)(rank + 1) * 2 + 0) * w + x];
}
// "some word of a[0 .. n) differs from b[0 .. n)": raises *flag (never clears it).  The halo rows a rank received against
// the ones it holds -- the exchange loop's stop test, left on the device so that it can be max-reduced there.
__global__ void k_rows_differ(const uint32_t *__restrict__ a, const uint32_t *__restrict__ b, size_t n, uint32_t *flag) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool d = i < n && a[i] != b[i];
  if (__builtin_amdgcn_ballot_w64(d) != 0ull && (threadIdx.x & 63) == 0) *flag = 1u;
}
}  // namespace

__global__ void k_iota(uint32_t *p, size_t n, uint32_t first) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = first + (uint32_t)i;
}

// p[i] = first + i: the colours of a contiguous range of the caller's seed list (ws_block_init wants them spelt out)
hipError_t block_iota(hipStream_t s, uint32_t *p, size_t n, uint32_t first) {
  if (n == 0) return hipSuccess;
  k_iota<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, n, first);
  return hipGetLastError();
}

hipError_t block_rows_differ(hipStream_t s, const uint32_t *a, const uint32_t *b, size_t n, uint32_t *flag) {
  if (n == 0) return hipSuccess;
  k_rows_differ<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, b, n, flag);
  return hipGetLastError();
}

// A block of a field cut in BOTH directions (ws_segment_tiled2d_device) has halo columns as well: column xl and column xr of
// a plane of pitch w as two contiguous runs of h words (what a neighbour wants), and such runs back into columns 0 / w - 1.
__global__ void k_pack_cols(const uint32_t *__restrict__ plane, size_t h, size_t w, size_t xl, size_t xr, uint32_t *out) {
  const size_t y = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= h) return;
  out[y] = plane[y * w + xl];
  out[h + y] = plane[y * w + xr];
}
__global__ void k_unpack_cols(uint32_t *plane, size_t h, size_t w, const uint32_t *__restrict__ left, const uint32_t *__restrict__ right) {
  const size_t y = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= h) return;
  if (left) plane[y * w] = left[y];
  if (right) plane[y * w + w - 1] = right[y];
}
hipError_t block_pack_cols(hipStream_t s, const uint32_t *plane, size_t h, size_t w, size_t xl, size_t xr, uint32_t *out) {
  if (h == 0 || w == 0) return hipSuccess;
  k_pack_cols<<<(unsigned)((h + 255) / 256), 256, 0, s>>>(plane, h, w, xl, xr, out);
  return hipGetLastError();
}
hipError_t block_unpack_cols(hipStream_t s, uint32_t *plane, size_t h, size_t w, const uint32_t *left, const uint32_t *right) {
  if (h == 0 || w == 0 || (!left && !right)) return hipSuccess;
  k_unpack_cols<<<(unsigned)((h + 255) / 256), 256, 0, s>>>(plane, h, w, left, right);
  return hipGetLastError();
}

hipError_t block_export_boundary(hipStream_t s, const uint32_t *labels, int h, int w, int halo_flags, uint32_t rank, uint32_t *rows) {
  if (w == 0 || h == 0) return hipSuccess;
  k_block_export<<<(unsigned)((2 * (size_t)w + 255) / 256), 256, 0, s>>>(labels, h, w, halo_flags, rank, rows);
  return hipGetLastError();
}

hipError_t block_import_boundary(hipStream_t s, const uint32_t *table, uint32_t world, uint32_t rank, uint32_t *resolved,
                                 uint32_t *labels, int h, int w, int halo_flags) {
  if (w == 0 || h == 0) return hipSuccess;
  const size_t n = (size_t)world * 2 * w;
  k_table_jump<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(table, n, resolved);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  k_fill_halo_rows<<<(unsigned)((w + 255) / 256), 256, 0, s>>>(resolved, rank, labels, h, w, halo_flags);
  return hipGetLastError();
}

}  // namespace wsk
