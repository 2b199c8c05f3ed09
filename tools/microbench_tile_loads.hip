// microbench_tile_loads.hip -- how fast can a workgroup-per-tile kernel move a u32 plane on MI355X?
// Build on the box: hipcc --offload-arch=gfx950 -O3 -o /tmp/mb tools/microbench_tile_loads.hip
// Variants: linear grid-stride copy (reference), tile pattern with PH rows per thread (256-wide
// tiles, 16 B per lane per row) at different launch bounds / dummy register pressure.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_copy(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, size_t n4) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (; i < n4; i += step) out[i] = in[i] + 1u;
}

// tile = 256 x (4*PH) pixels, thread = 4 x PH patch; read plane `in`, write plane `out`
template <int PH, int MINW>
__global__ __launch_bounds__(256, MINW) void k_tile(const unsigned *__restrict__ in, unsigned *__restrict__ out, int H, int W,
                                                   int tilesX, int do_store, int spin) {
  const int tile_x = blockIdx.x % tilesX, tile_y = blockIdx.x / tilesX;
  const int lane = threadIdx.x & 63, band = threadIdx.x >> 6;
  const int gx0 = tile_x * 256 + lane * 4, gyb = tile_y * 4 * PH + band * PH;
  u32x4 v[PH];
#pragma unroll
  for (int r = 0; r < PH; ++r) v[r] = *reinterpret_cast<const u32x4 *>(in + (size_t)(gyb + r) * W + gx0);
  unsigned acc = 0;
  for (int s = 0; s < spin; ++s) {
#pragma unroll
    for (int r = 0; r < PH; ++r) { v[r] = v[r] * 3u + 1u; acc += v[r].x; }
  }
  if (do_store) {
#pragma unroll
    for (int r = 0; r < PH; ++r) *reinterpret_cast<u32x4 *>(out + (size_t)(gyb + r) * W + gx0) = v[r] + acc;
  } else {
    unsigned s = acc;
#pragma unroll
    for (int r = 0; r < PH; ++r) s += v[r].x + v[r].y + v[r].z + v[r].w;
    if (s == 0x12345678u) out[0] = s;
  }
}

template <class F>
float time_ms(F f, int reps = 10) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); f();
  CHECK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms; CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int H = 8192, W = 8192;
  const size_t n = (size_t)H * W;
  unsigned *in, *out;
  CHECK(hipMalloc(&in, n * 4)); CHECK(hipMalloc(&out, n * 4));
  CHECK(hipMemset(in, 1, n * 4)); CHECK(hipMemset(out, 0, n * 4));
  const double gb = n * 4 / 1e9;
  {
    float ms = time_ms([&] { k_copy<<<16384, 256>>>((const u32x4 *)in, (u32x4 *)out, n / 4); });
    printf("linear copy           : %.3f ms  read %.2f TB/s (+ same written)\n", ms, gb / ms);
  }
#define RUN(PH, MINW, STORE, SPIN)                                                                         \
  {                                                                                                        \
    const int tx = W / 256, ty = H / (4 * PH);                                                             \
    float ms = time_ms([&] { k_tile<PH, MINW><<<tx * ty, 256>>>(in, out, H, W, tx, STORE, SPIN); });       \
    printf("tile PH=%2d minw=%d store=%d spin=%3d : %.3f ms  read %.2f TB/s\n", PH, MINW, STORE, SPIN, ms, gb / ms); \
  }
  RUN(8, 1, 0, 0) RUN(8, 1, 1, 0) RUN(8, 1, 1, 20) RUN(8, 1, 1, 100)
  RUN(4, 1, 0, 0) RUN(4, 1, 1, 0)
  RUN(16, 1, 0, 0) RUN(16, 1, 1, 0)
  RUN(2, 1, 0, 0) RUN(2, 1, 1, 0)
  return 0;
}
