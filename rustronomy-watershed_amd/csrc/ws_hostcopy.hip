// Labels to a host caller's u64 plane (Array2<usize>, lib.rs:1663-1666) without sending u64 over PCIe.
//
// The device holds 4-byte labels; the reference's callers want 8-byte ones.  Widening on the device and copying 8 bytes a
// pixel makes the copy the whole cost of a host call (8192^2: 512 MiB at ~52 GB/s = 10 ms of an 11.5 ms call).  Here the
// 4-byte plane crosses the bus in chunks into pinned staging slots and host threads widen every chunk into the caller's
// memory while the next ones are in flight: the bus carries half the bytes, the widening hides behind it.
//
//   producer (the calling thread)   issues chunk copies while a staging slot is free, waits for the oldest copy's event,
//                                   publishes the chunk (an atomic count; nobody sleeps)
//   T workers                       take every published chunk in order, each its own contiguous 1/T of it, and store
//                                   with non-temporal 16-byte stores (no read-for-ownership of the caller's lines)
//
// Not a compute path: no label is decided here.
#include "ws_ctx.hpp"

#include <emmintrin.h>

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <memory>
#include <system_error>
#include <thread>

namespace wsapi {

namespace {

constexpr size_t HC_CHUNK_MAX = (size_t)1 << 22;      // labels per chunk: 16 MiB over the bus, 32 MiB written
constexpr size_t HC_CHUNK_MIN = (size_t)1 << 19;
constexpr int HC_SLOTS = 4;
// planes below this take the one-copy path: starting the threads costs more than they hide (ws_segment_minima, u64 labels,
// one copy / chunks: 1024^2 0.32 / 0.40 ms, 1448^2 0.52 / 0.54, 2048^2 0.84 / 0.74-0.79, 4096^2 2.9 / 2.1, 8192^2 11.5 / 7.1)
constexpr size_t HC_MIN = (size_t)1 << 21;

// a quarter of the plane, so that planes of a few chunks' worth still overlap their copies with the widening (2048^2: one chunk
// of 2^22 0.92 ms, four of 2^20 0.74), between 2^19 and 2^22 labels
size_t chunk_of(size_t n) {
  size_t ch = std::min(HC_CHUNK_MAX, std::max(HC_CHUNK_MIN, ((n + 3) / 4 + 4095) & ~(size_t)4095));
  if (const char *e = tuning_env("WS_HOST_CHUNK_LOG2")) ch = std::min(HC_CHUNK_MAX, (size_t)1 << atoi(e));      // A/B knob, tools/ only
  return ch;
}

// threads of ws_ctx_set_host_threads (default 4: 2 .. 8 threads and chunks of 2^21 .. 2^23 labels all take 7.1-7.4 ms at 8192^2),
// never more than the machine has
int host_threads(const ws_ctx *c) {
  const unsigned hw = std::thread::hardware_concurrency();
  int t = std::min(c->host_threads, (int)(hw ? hw : 4u));
  if (const char *e = tuning_env("WS_HOST_THREADS")) t = std::max(0, std::min(64, std::atoi(e)));
  return t;
}

void widen_span(const uint32_t *__restrict__ src, uint64_t *__restrict__ dst, size_t n) {
  size_t i = 0;
  for (; i < n && (reinterpret_cast<uintptr_t>(dst + i) & 15u); ++i) dst[i] = src[i];
  const __m128i zero = _mm_setzero_si128();
  for (; i + 4 <= n; i += 4) {
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i *>(src + i));
    _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i), _mm_unpacklo_epi32(v, zero));
    _mm_stream_si128(reinterpret_cast<__m128i *>(dst + i + 2), _mm_unpackhi_epi32(v, zero));
  }
  for (; i < n; ++i) dst[i] = src[i];
  _mm_sfence();
}

// Waits are spins (pause, then yield): a chunk is ~150 us of bus time, and a sleeping thread's wake-up (~50 us with a condition
// variable, more with many sleepers) was a fifth of the call -- ws_segment_minima at 8192^2, 16 threads on condition
// variables: 8.2 ms, 4: 7.5; spinning: 7.1; one plain 8-byte copy: 11.5; u32 labels: 6.5 (tools/ab_hostcopy.sh, tuning build).
template <class Pred>
void spin_until(Pred pred) {
  for (unsigned it = 0; !pred(); ++it) {
    if (it < 4096) _mm_pause();
    else std::this_thread::yield();
  }
}

}  // namespace

bool host_copy_in_chunks(const ws_ctx *c, size_t n) { return n >= HC_MIN && host_threads(c) > 0; }

void host_copy_release(ws_ctx *c) {
  if (c->hc_stage) (void)hipHostFree(c->hc_stage);
  c->hc_stage = nullptr;
  for (auto &e : c->hc_ev)
    if (e) { (void)hipEventDestroy(e); e = nullptr; }
}

static int host_copy_slots(ws_ctx *c) {
  if (c->hc_stage) return WS_OK;
  HIP_TRY(c, hipHostMalloc((void **)&c->hc_stage, HC_SLOTS * HC_CHUNK_MAX * sizeof(uint32_t), hipHostMallocDefault));
  for (auto &e : c->hc_ev)
    if (!e) HIP_TRY(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return WS_OK;
}

// the plane widened on the device and copied whole (small planes, no host threads)
static int one_copy(ws_ctx *c, const uint32_t *d_labels, uint64_t *out, size_t n, hipStream_t st) {
  // (callers that park their u32 words in the context's u64 buffer must keep to the chunked road -- host_copy_in_chunks: this
  // path widens INTO that buffer and may reallocate it)
  if (c->out64.p && (const char *)d_labels >= (const char *)c->out64.p && (const char *)d_labels < (const char *)c->out64.p + c->out64.cap)
    return fail(c, WS_ERR_BAD_ARG, "internal: the words to widen sit in the buffer they would be widened into");
  int rc = ensure(c, c->out64, std::max<size_t>(n, 1) * sizeof(uint64_t));
  if (rc) return rc;
  HIP_TRY(c, widen_labels(st, d_labels, (uint64_t *)c->out64.p, n));
  HIP_TRY(c, hipMemcpyAsync(out, c->out64.p, n * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  return WS_OK;
}

int labels_to_host_u64(ws_ctx *c, const uint32_t *d_labels, uint64_t *out, size_t n, hipStream_t on, size_t row_len, size_t out_pitch) {
  const hipStream_t st = on ? on : c->stream;
  const int T = host_threads(c);
  // (row_len != 0: the n words are rows of row_len that go to rows of out_pitch words in the caller's plane -- a tile's rectangle)
  if (row_len != 0 && (!host_copy_in_chunks(c, n) || n % row_len != 0)) return fail(c, WS_ERR_BAD_ARG, "internal: a rectangle takes the chunked road or none");
  if (!host_copy_in_chunks(c, n)) return one_copy(c, d_labels, out, n, st);
  // words [k0, k0 + len) of the packed source into their places in the caller's memory
  auto widen_to = [&](const uint32_t *src, size_t k0, size_t len) {
    if (row_len == 0) { widen_span(src, out + k0, len); return; }
    while (len) {
      const size_t r = k0 / row_len, col = k0 % row_len, run = std::min(len, row_len - col);
      widen_span(src, out + r * out_pitch + col, run);
      src += run; k0 += run; len -= run;
    }
  };
  if (int rc = host_copy_slots(c)) return rc;
  const size_t HC_CHUNK = chunk_of(n);
  const size_t nch = (n + HC_CHUNK - 1) / HC_CHUNK;
  std::atomic<size_t> published{0};      // chunks whose copy has landed in its slot
  std::atomic<bool> abort{false};
  std::unique_ptr<std::atomic<int>[]> done(new std::atomic<int>[nch]);      // workers finished with chunk i
  for (size_t i = 0; i < nch; ++i) done[i].store(0, std::memory_order_relaxed);
  uint32_t *stage = c->hc_stage;
  auto worker = [&](int t) {
    for (size_t i = 0; i < nch; ++i) {
      spin_until([&] { return published.load(std::memory_order_acquire) > i || abort.load(std::memory_order_relaxed); });
      if (abort.load(std::memory_order_relaxed)) return;
      const size_t len = std::min(HC_CHUNK, n - i * HC_CHUNK);
      const size_t a = len * (size_t)t / (size_t)T, b = len * (size_t)(t + 1) / (size_t)T;
      widen_to(stage + (i % HC_SLOTS) * HC_CHUNK + a, i * HC_CHUNK + a, b - a);
      done[i].fetch_add(1, std::memory_order_release);
    }
  };
  std::vector<std::thread> pool;
  pool.reserve(T);
  try {
    if (tuning_env("WS_HOST_THREADS_FAIL")) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));      // (tuning build: rehearses the catch)
    for (int t = 0; t < T; ++t) pool.emplace_back(worker, t);
  } catch (const std::system_error &) {      // the process may not start (more) threads: nothing has been published yet
    abort.store(true);
    for (auto &th : pool) th.join();
    // the calling thread widens by itself, chunk after chunk (not one_copy: the source may live in the context's u64 buffer)
    for (size_t i = 0; i < nch; ++i) {
      const size_t len = std::min(HC_CHUNK, n - i * HC_CHUNK);
      HIP_TRY(c, hipMemcpyAsync(stage, d_labels + i * HC_CHUNK, len * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      HIP_TRY(c, hipStreamSynchronize(st));
      widen_to(stage, i * HC_CHUNK, len);
    }
    return WS_OK;
  }
  hipError_t err = hipSuccess;
  size_t issued = 0, landed = 0, freed = 0;      // freed: chunks every worker has finished (workers take chunks in order)
  while (landed < nch && err == hipSuccess) {
    while (freed < landed && done[freed].load(std::memory_order_acquire) == T) ++freed;
    while (issued < nch && issued - freed < (size_t)HC_SLOTS && err == hipSuccess) {
      const size_t len = std::min(HC_CHUNK, n - issued * HC_CHUNK);
      err = hipMemcpyAsync(stage + (issued % HC_SLOTS) * HC_CHUNK, d_labels + issued * HC_CHUNK, len * sizeof(uint32_t),
                           hipMemcpyDeviceToHost, st);
      if (err == hipSuccess) err = hipEventRecord(c->hc_ev[issued % HC_SLOTS], st);
      ++issued;
    }
    if (err != hipSuccess) break;
    if (landed < issued) {
      err = hipEventSynchronize(c->hc_ev[landed % HC_SLOTS]);
      if (err != hipSuccess) break;
      published.store(++landed, std::memory_order_release);
    } else {      // every slot holds a chunk that is still being widened
      spin_until([&] { return done[freed].load(std::memory_order_acquire) == T; });
    }
  }
  if (err != hipSuccess) abort.store(true);
  for (auto &th : pool) th.join();
  if (err != hipSuccess) {
    (void)hipStreamSynchronize(st);      // no copy may still be writing a slot when the context goes on
    HIP_TRY(c, err);
  }
  return WS_OK;
}

// (The other direction was tried too: host threads narrowing the u64 seed pairs into the pinned slots, 8 bytes a seed over the
// bus instead of 16.  Four threads narrow 7.3 M pairs in ~4 ms, the bus carries the 16-byte pairs in 2.2: ws_segment at 8192^2
// 10.0 -> 11.7 ms.  The pairs are copied whole and narrowed by k_narrow_seeds.)

}  // namespace wsapi
