// ws_relax2.hip -- second-generation relaxation kernel of the fused engine (gfx950, wave64).
//
// Same fixpoint as k_relax (ws_kernels.hip): key(p) = max(base(p), 1 + min4 key(q)).  What changes
// is where the stamps live while a tile iterates:
//
//   * every thread owns a 4-wide x PH-tall patch of pixels IN REGISTERS (stamps + bases) and
//     sweeps it Gauss-Seidel fashion, forward then backward: inside a patch information travels
//     for free;
//   * a wave is one 256-pixel-wide band (64 lanes x 4 px); the left/right patch columns come from
//     the neighbouring LANES with one DPP wave shift each (v_mov_b32_dpp wave_shr:1 / wave_shl:1,
//     the `old` operand supplies the tile's halo column for lane 0 / 63) -- no LDS, no conflicts;
//   * the 4 waves of a workgroup are 4 bands stacked vertically (tile = 256 x 4*PH px); only the
//     band boundary rows go through LDS, as one ds_write_b128 / ds_read_b128 per lane per row;
//   * global loads and stores are 16 B per lane, 1 KiB per wave instruction, row contiguous.
//
// Per outer iteration a thread does 2 x 4*PH pixel updates of ~7 VALU ops each, 4*PH DPP moves,
// 2 LDS reads and 2 LDS writes; convergence is one workgroup OR per iteration.
#include "ws_common.hpp"

namespace wsk {

constexpr int R2_PW = 4;            // patch width (pixels per lane along x)
constexpr int R2_NW = 4;            // waves (bands) per workgroup
constexpr int R2_TW = 64 * R2_PW;   // tile width: 256

template <int DPP_CTRL>
__device__ __forceinline__ uint32_t dpp_shift(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, DPP_CTRL, 0xF, 0xF, false);
}
// lane i <- lane i-1 (lane 0 keeps `old`)
__device__ __forceinline__ uint32_t from_left_lane(uint32_t old, uint32_t v) { return dpp_shift<0x138>(old, v); }   // wave_shr:1
// lane i <- lane i+1 (lane 63 keeps `old`)
__device__ __forceinline__ uint32_t from_right_lane(uint32_t old, uint32_t v) { return dpp_shift<0x130>(old, v); }  // wave_shl:1

template <int PH>
__global__ __launch_bounds__(256) void k_relax2(const uint8_t *__restrict__ img, size_t img_stride, uint32_t *keys,
                                                int H, int W, int tilesX, int tilesY, uint32_t max_level,
                                                uint32_t pass, uint32_t *stamps, uint32_t *counters,
                                                uint32_t *overflow, uint32_t *tiles_run, uint32_t *any_change) {
  constexpr int TH = R2_NW * PH;
  // row 0: halo above the tile; rows 1+2w / 2+2w: top / bottom row of band w; last: halo below
  __shared__ __attribute__((aligned(16))) uint32_t sRow[2 * R2_NW + 2][R2_TW];
  __shared__ uint32_t s_edges;

  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const size_t ntiles = (size_t)tilesX * tilesY;
  const uint32_t *stamps_prev = stamps + ((pass + 1) & 1) * ntiles * 4;
  uint32_t *stamps_cur = stamps + (pass & 1) * ntiles * 4;
  if (blockIdx.x == 0 && threadIdx.x == 0) counters[(pass + 1) % COUNTER_RING] = 0;
  if (pass != 0) {
    const int t = tile_y * tilesX + tile_x;
    bool run = false;
    if (tile_y > 0) run |= stamps_prev[(size_t)(t - tilesX) * 4 + 1] == pass;
    if (tile_y + 1 < tilesY) run |= stamps_prev[(size_t)(t + tilesX) * 4 + 0] == pass;
    if (tile_x > 0) run |= stamps_prev[(size_t)(t - 1) * 4 + 3] == pass;
    if (tile_x + 1 < tilesX) run |= stamps_prev[(size_t)(t + 1) * 4 + 2] == pass;
    if (!run) return;
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, band = tid >> 6;
  const int x0 = tile_x * R2_TW, y0 = tile_y * TH;
  const int gx0 = x0 + lane * R2_PW;          // first column of this lane's patch
  const int gyb = y0 + band * PH;             // first row of this band
  if (tid == 0) { s_edges = 0; atomicAdd(tiles_run, 1u); }

  uint32_t T[PH][R2_PW], B[PH][R2_PW];
  uint32_t haloL[PH], haloR[PH];
  const bool full_x = gx0 + R2_PW <= W;

#pragma unroll
  for (int r = 0; r < PH; ++r) {
    const int gy = gyb + r;
    const bool row_ok = gy < H;
    // stamps: one 16-byte load per lane where the whole patch row is inside the image
    if (row_ok && full_x) {
      const uint4 v = *reinterpret_cast<const uint4 *>(keys + (size_t)gy * W + gx0);
      T[r][0] = v.x; T[r][1] = v.y; T[r][2] = v.z; T[r][3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < R2_PW; ++c) T[r][c] = (row_ok && gx0 + c < W) ? keys[(size_t)gy * W + gx0 + c] : KEY_INF;
    }
    // bases: only interior pixels with img <= max level can ever be flooded (lib.rs:220-224)
    const bool row_int = gy >= 1 && gy < H - 1;
#pragma unroll
    for (int c = 0; c < R2_PW; ++c) {
      const int gx = gx0 + c;
      uint32_t b = KEY_INF;
      if (row_int && gx >= 1 && gx < W - 1) {
        const uint32_t v = img[(size_t)gy * img_stride + gx];
        if (v <= max_level) b = (v << 24) | 1u;
      }
      B[r][c] = b;
    }
    // tile halo columns (only lane 0 / lane 63 ever use theirs)
    haloL[r] = (lane == 0 && row_ok && x0 > 0) ? keys[(size_t)gy * W + x0 - 1] : KEY_INF;
    haloR[r] = (lane == 63 && row_ok && x0 + R2_TW < W) ? keys[(size_t)gy * W + x0 + R2_TW] : KEY_INF;
  }

  // halo rows above / below the tile, and this band's boundary rows, into LDS
  {
    uint4 v = make_uint4(KEY_INF, KEY_INF, KEY_INF, KEY_INF);
    if (band == 0 || band == R2_NW - 1) {
      const int gy = band == 0 ? y0 - 1 : y0 + TH;
      const bool top_only = band == 0;
      if (gy >= 0 && gy < H) {
        if (full_x) v = *reinterpret_cast<const uint4 *>(keys + (size_t)gy * W + gx0);
        else {
          if (gx0 + 0 < W) v.x = keys[(size_t)gy * W + gx0 + 0];
          if (gx0 + 1 < W) v.y = keys[(size_t)gy * W + gx0 + 1];
          if (gx0 + 2 < W) v.z = keys[(size_t)gy * W + gx0 + 2];
          if (gx0 + 3 < W) v.w = keys[(size_t)gy * W + gx0 + 3];
        }
      }
      *reinterpret_cast<uint4 *>(&sRow[top_only ? 0 : 2 * R2_NW + 1][lane * R2_PW]) = v;
      if (R2_NW == 1) {   // single band: it owns both halo rows (not used with R2_NW = 4)
      }
    }
    *reinterpret_cast<uint4 *>(&sRow[1 + 2 * band][lane * R2_PW]) = make_uint4(T[0][0], T[0][1], T[0][2], T[0][3]);
    *reinterpret_cast<uint4 *>(&sRow[2 + 2 * band][lane * R2_PW]) =
        make_uint4(T[PH - 1][0], T[PH - 1][1], T[PH - 1][2], T[PH - 1][3]);
  }
  __syncthreads();

  bool row_changed[PH];
#pragma unroll
  for (int r = 0; r < PH; ++r) row_changed[r] = false;
  bool left_changed = false, right_changed = false;

  for (;;) {
    bool changed = false;
    const uint4 up4 = *reinterpret_cast<const uint4 *>(&sRow[2 * band][lane * R2_PW]);
    const uint4 dn4 = *reinterpret_cast<const uint4 *>(&sRow[2 * band + 3][lane * R2_PW]);
    const uint32_t up[R2_PW] = {up4.x, up4.y, up4.z, up4.w};
    const uint32_t dn[R2_PW] = {dn4.x, dn4.y, dn4.z, dn4.w};
    uint32_t L[PH], R[PH];

    // ---- forward sweep (top-left to bottom-right) ----
#pragma unroll
    for (int r = 0; r < PH; ++r) { L[r] = from_left_lane(haloL[r], T[r][R2_PW - 1]); R[r] = from_right_lane(haloR[r], T[r][0]); }
#pragma unroll
    for (int r = 0; r < PH; ++r) {
#pragma unroll
      for (int c = 0; c < R2_PW; ++c) {
        const uint32_t u = r == 0 ? up[c] : T[r - 1][c];
        const uint32_t d = r == PH - 1 ? dn[c] : T[r + 1][c];
        const uint32_t l = c == 0 ? L[r] : T[r][c - 1];
        const uint32_t rt = c == R2_PW - 1 ? R[r] : T[r][c + 1];
        const uint32_t cand = max(B[r][c], min(min(u, d), min(l, rt)) + 1u);
        const bool lower = cand < T[r][c];
        T[r][c] = min(T[r][c], cand);
        changed |= lower;
        row_changed[r] |= lower;
        if (c == 0) left_changed |= lower;
        if (c == R2_PW - 1) right_changed |= lower;
      }
    }
    // ---- backward sweep (bottom-right to top-left), with refreshed lane neighbours ----
#pragma unroll
    for (int r = 0; r < PH; ++r) { L[r] = from_left_lane(haloL[r], T[r][R2_PW - 1]); R[r] = from_right_lane(haloR[r], T[r][0]); }
#pragma unroll
    for (int r = PH - 1; r >= 0; --r) {
#pragma unroll
      for (int c = R2_PW - 1; c >= 0; --c) {
        const uint32_t u = r == 0 ? up[c] : T[r - 1][c];
        const uint32_t d = r == PH - 1 ? dn[c] : T[r + 1][c];
        const uint32_t l = c == 0 ? L[r] : T[r][c - 1];
        const uint32_t rt = c == R2_PW - 1 ? R[r] : T[r][c + 1];
        const uint32_t cand = max(B[r][c], min(min(u, d), min(l, rt)) + 1u);
        const bool lower = cand < T[r][c];
        T[r][c] = min(T[r][c], cand);
        changed |= lower;
        row_changed[r] |= lower;
        if (c == 0) left_changed |= lower;
        if (c == R2_PW - 1) right_changed |= lower;
      }
    }
    // publish this band's boundary rows for the bands above / below
    *reinterpret_cast<uint4 *>(&sRow[1 + 2 * band][lane * R2_PW]) = make_uint4(T[0][0], T[0][1], T[0][2], T[0][3]);
    *reinterpret_cast<uint4 *>(&sRow[2 + 2 * band][lane * R2_PW]) =
        make_uint4(T[PH - 1][0], T[PH - 1][1], T[PH - 1][2], T[PH - 1][3]);
    if (!__syncthreads_or(changed ? 1 : 0)) break;
  }

  // write back the rows that changed (16 B per lane), collect edge flags and the ring-carry check
  uint32_t e = 0, ovf = 0;
#pragma unroll
  for (int r = 0; r < PH; ++r) {
    const int gy = gyb + r;
    if (row_changed[r] && gy < H) {
      if (full_x) {
        *reinterpret_cast<uint4 *>(keys + (size_t)gy * W + gx0) = make_uint4(T[r][0], T[r][1], T[r][2], T[r][3]);
      } else {
#pragma unroll
        for (int c = 0; c < R2_PW; ++c) if (gx0 + c < W) keys[(size_t)gy * W + gx0 + c] = T[r][c];
      }
#pragma unroll
      for (int c = 0; c < R2_PW; ++c)
        ovf |= (T[r][c] != 0u && T[r][c] < KEY_INF && (T[r][c] & RING_MASK) == 0u);   // ring field carried into the level
      e |= 16u;
    }
  }
  if (ovf) atomicExch(overflow, 1u);
  if (band == 0 && row_changed[0]) e |= 1u;
  if (band == R2_NW - 1 && row_changed[PH - 1]) e |= 2u;
  if (lane == 0 && left_changed) e |= 4u;
  if (lane == 63 && right_changed) e |= 8u;
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
      if (ed & 15u) atomicAdd(&counters[pass % COUNTER_RING], 1u);
      atomicExch(any_change, 1u);
    }
  }
}

size_t relax2_tiles(int h, int w, int ph) {
  const int th = R2_NW * ph;
  return (size_t)((w + R2_TW - 1) / R2_TW) * ((h + th - 1) / th);
}

hipError_t relax2_pass(hipStream_t s, int ph, const uint8_t *img, size_t img_stride, uint32_t *keys, int h, int w,
                       uint32_t max_level, uint32_t pass, uint32_t *stamps, uint32_t *counters, uint32_t *overflow,
                       uint32_t *tiles_run, uint32_t *any_change) {
  const int th = R2_NW * ph;
  const int tx = (w + R2_TW - 1) / R2_TW, ty = (h + th - 1) / th;
  if (ph == 16)
    k_relax2<16><<<tx * ty, 256, 0, s>>>(img, img_stride, keys, h, w, tx, ty, max_level, pass, stamps, counters, overflow,
                                         tiles_run, any_change);
  else
    k_relax2<8><<<tx * ty, 256, 0, s>>>(img, img_stride, keys, h, w, tx, ty, max_level, pass, stamps, counters, overflow,
                                        tiles_run, any_change);
  return hipGetLastError();
}

}  // namespace wsk
