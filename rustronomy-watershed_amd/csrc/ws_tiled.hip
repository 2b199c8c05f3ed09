// ws_tiled.hip -- several GPUs behind one call: ws_group_*, ws_segment_tiled(_device), ws_segment_batch_group.
//
// The reference's drivers are ONE address space (lib.rs:1689-1748: a rayon loop over the windows of one array); what
// stands in for that across devices is a group of RANKS, each a ws_ctx on one device, running the row-block steps of
// ws_block_api.hip (the rank program below is the loop of distributed.py, in C++) around three exchange steps:
//
//   swap      first / last OWNED row of a plane -> the neighbour's receive buffer (the plane itself is not written: the
//             caller compares first, on the device)
//   reduce    max or min of one u32 word over the ranks -> a host value on every rank (ONE host read)
//   gather    `words` u32 of every rank -> a table of world * words on every rank
//
// Two implementations (struct Exchange): LOCAL -- the ranks are host threads of this process, the steps are stream-ordered
// copies between their buffers behind a thread barrier (peer-to-peer when the ranks sit on different devices) -- and RCCL
// -- this process is one rank; grouped ncclSend / ncclRecv, ncclAllReduce, ncclAllGather on the rank's stream over xGMI.
// librccl.so is loaded with dlopen when the first RCCL group is made.
#include "ws_ctx.hpp"

#include <rccl/rccl.h>

#include <dlfcn.h>

#include <condition_variable>
#include <mutex>
#include <thread>

using namespace wsapi;

namespace {

// ---- RCCL, resolved at run time -----------------------------------------------------------------------------------------
struct RcclApi {
  void *lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  std::string why;
};

RcclApi *rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) { api.why = std::string("dlopen(librccl.so): ") + (dlerror() ? dlerror() : "not found"); return; }
    bool ok = true;
    auto sym = [&](const char *n) { void *p = dlsym(api.lib, n); if (!p) { ok = false; api.why = std::string("librccl.so lacks ") + n; } return p; };
    api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
    api.CommAbort = (decltype(api.CommAbort))sym("ncclCommAbort");
    api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
    api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
    api.Send = (decltype(api.Send))sym("ncclSend");
    api.Recv = (decltype(api.Recv))sym("ncclRecv");
    api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))sym("ncclAllGather");
    if (!ok) { dlclose(api.lib); api.lib = nullptr; }
  });
  return &api;
}

// ---- a thread barrier that can be broken (a rank that fails must not leave the others waiting) ------------------------------
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 1, count = 0;
  uint64_t gen = 0;
  bool broken = false;
  bool wait() {
    std::unique_lock<std::mutex> l(m);
    if (broken) return false;
    const uint64_t g = gen;
    if (++count == n) { count = 0; ++gen; cv.notify_all(); return true; }
    cv.wait(l, [&] { return gen != g || broken; });
    return !broken;
  }
  void abort() { std::lock_guard<std::mutex> l(m); broken = true; cv.notify_all(); }
  void reset(int ranks) { std::lock_guard<std::mutex> l(m); n = ranks; count = 0; broken = false; }
};

struct Grow {      // a device buffer that only ever grows
  void *p = nullptr;
  size_t cap = 0;
};

// one rank driven by this process
struct Rank {
  int rank = 0, device = 0;
  ws_ctx *ctx = nullptr;
  Grow keys, labels, recv, rows, table, parent, img, seeds, colours, out64, cols, full_keys, full_labels;
  uint32_t *flag = nullptr;           // device: 4 words (the exchange loop's stop word; reduce scratch)
  uint32_t *flag_host = nullptr;      // pinned mirror
  // what the LOCAL exchange steps read from their neighbours (published before the barrier)
  const uint32_t *pub_first = nullptr, *pub_last = nullptr, *pub_send = nullptr, *pub_cols = nullptr;
  uint32_t pub_word = 0;
};

}  // namespace

struct ws_group {
  bool is_rccl = false;
  int world = 1, first_local = 0;
  std::vector<Rank> ranks;      // the local ones
  Barrier barrier;
  ncclComm_t comm = nullptr;
  std::mutex err_m;
  std::string err;
};

namespace {

int gfail(ws_group *g, int code, const std::string &what) {
  if (g) { std::lock_guard<std::mutex> l(g->err_m); if (g->err.empty() || code != WS_ERR_HIP) g->err = what; }
  return code;
}

#define G_HIP(g, call)                                                                                          \
  do {                                                                                                          \
    hipError_t e_ = (call);                                                                                     \
    if (e_ != hipSuccess) return gfail((g), e_ == hipErrorOutOfMemory ? WS_ERR_OOM : WS_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define G_NCCL(g, call)                                                                                         \
  do {                                                                                                          \
    ncclResult_t r_ = (call);                                                                                   \
    if (r_ != ncclSuccess) return gfail((g), WS_ERR_RCCL, std::string(#call) + ": " + rccl()->GetErrorString(r_)); \
  } while (0)
// a ws_* call on the rank's own context: its message travels to the group
#define G_WS(g, rk, call)                                                                                       \
  do {                                                                                                          \
    const int rc_ = (call);                                                                                     \
    if (rc_ != WS_OK) return gfail((g), rc_, std::string("rank ") + std::to_string((rk).rank) + ": " + #call + ": " + ws_last_error((rk).ctx)); \
  } while (0)

// the status of a ws_* call on a rank's own context, its message carried to the group (WS_ERR_CAPACITY included: the caller reads n_lakes)
int gfail_if(ws_group *g, const Rank &rk, int rc) {
  return rc == WS_OK ? WS_OK : gfail(g, rc, std::string("rank ") + std::to_string(rk.rank) + ": " + ws_last_error(rk.ctx));
}

int grow(ws_group *g, Grow &b, size_t bytes) {
  if (bytes <= b.cap) return WS_OK;
  if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
  const size_t want = bytes + (bytes >> 4) + 256;
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) { b.p = nullptr; return gfail(g, WS_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  b.cap = want;
  return WS_OK;
}

// ---- the three exchange steps ------------------------------------------------------------------------------------------
struct Exchange {
  ws_group *g;
  Rank &me;
  hipStream_t stream() const { return me.ctx->stream; }

  Rank *local(int rank) const { return &g->ranks[(size_t)(rank - g->first_local)]; }

  // first / last owned row of `plane` (h x w; halo rows where the rank has neighbours) -> the neighbours' receive buffers:
  // me.recv[0 .. w) = the upper neighbour's last owned row, me.recv[w .. 2w) = the lower neighbour's first owned row
  int swap(const uint32_t *plane, size_t h, size_t w) {
    const bool up = me.rank > 0, down = me.rank < g->world - 1;
    const uint32_t *first = plane + (up ? w : 0), *last = plane + (h - 1 - (down ? 1 : 0)) * w;
    uint32_t *recv = (uint32_t *)me.recv.p;
    if (g->is_rccl) {
      RcclApi *n = rccl();
      G_NCCL(g, n->GroupStart());
      if (up) { G_NCCL(g, n->Send(first, w, ncclUint32, me.rank - 1, g->comm, stream())); G_NCCL(g, n->Recv(recv, w, ncclUint32, me.rank - 1, g->comm, stream())); }
      if (down) { G_NCCL(g, n->Send(last, w, ncclUint32, me.rank + 1, g->comm, stream())); G_NCCL(g, n->Recv(recv + w, w, ncclUint32, me.rank + 1, g->comm, stream())); }
      G_NCCL(g, n->GroupEnd());
      return WS_OK;
    }
    // (every rank's plane is complete in memory: the block steps end with a wait for their stream)
    me.pub_first = first; me.pub_last = last;
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    if (up) G_HIP(g, hipMemcpyAsync(recv, local(me.rank - 1)->pub_last, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    if (down) G_HIP(g, hipMemcpyAsync(recv + w, local(me.rank + 1)->pub_first, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    G_HIP(g, hipStreamSynchronize(stream()));
    // nobody goes on to rewrite its plane while a neighbour still reads it
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    return WS_OK;
  }

  // A block of a field cut in both directions: nb[] = the ranks above, below, left and right of me (-1: the field's edge).
  // Sends my first / last OWNED row and column, receives the neighbours' into me.recv: [0, w) from above, [w, 2w) from below,
  // [2w, 2w + h) from the left, [2w + h, 2w + 2h) from the right.  (Rows are contiguous; columns are packed into me.cols
  // first.  The four corner cells of the halo ring are nobody's neighbours in a 4-connected stencil.)
  int swap2d(const uint32_t *plane, size_t h, size_t w, const int nb[4]) {
    const bool up = nb[0] >= 0, down = nb[1] >= 0, left = nb[2] >= 0, right = nb[3] >= 0;
    const uint32_t *first = plane + (up ? w : 0), *last = plane + (h - 1 - (down ? 1 : 0)) * w;
    uint32_t *recv = (uint32_t *)me.recv.p, *cols = (uint32_t *)me.cols.p;
    G_HIP(g, block_pack_cols(stream(), plane, h, w, left ? 1 : 0, w - 1 - (right ? 1 : 0), cols));
    if (g->is_rccl) {
      RcclApi *n = rccl();
      G_NCCL(g, n->GroupStart());
      if (up) { G_NCCL(g, n->Send(first, w, ncclUint32, nb[0], g->comm, stream())); G_NCCL(g, n->Recv(recv, w, ncclUint32, nb[0], g->comm, stream())); }
      if (down) { G_NCCL(g, n->Send(last, w, ncclUint32, nb[1], g->comm, stream())); G_NCCL(g, n->Recv(recv + w, w, ncclUint32, nb[1], g->comm, stream())); }
      if (left) { G_NCCL(g, n->Send(cols, h, ncclUint32, nb[2], g->comm, stream())); G_NCCL(g, n->Recv(recv + 2 * w, h, ncclUint32, nb[2], g->comm, stream())); }
      if (right) { G_NCCL(g, n->Send(cols + h, h, ncclUint32, nb[3], g->comm, stream())); G_NCCL(g, n->Recv(recv + 2 * w + h, h, ncclUint32, nb[3], g->comm, stream())); }
      G_NCCL(g, n->GroupEnd());
      return WS_OK;
    }
    G_HIP(g, hipStreamSynchronize(stream()));      // my packed columns are complete before anyone copies them
    me.pub_first = first; me.pub_last = last; me.pub_cols = cols;
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    if (up) G_HIP(g, hipMemcpyAsync(recv, local(nb[0])->pub_last, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    if (down) G_HIP(g, hipMemcpyAsync(recv + w, local(nb[1])->pub_first, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    // (tiles of one tile row have the same rows: the neighbour's packed runs are h words long, its LAST column first comes second)
    if (left) G_HIP(g, hipMemcpyAsync(recv + 2 * w, local(nb[2])->pub_cols + h, h * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    if (right) G_HIP(g, hipMemcpyAsync(recv + 2 * w + h, local(nb[3])->pub_cols, h * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    G_HIP(g, hipStreamSynchronize(stream()));
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    return WS_OK;
  }

  // every rank's OWNED rows of a plane (`own`: its first owned row) -> rank 0's whole plane of field_h x w words (`full`, rank 0
  // only): one message per rank and plane.  Returns with rank 0's copies complete.
  int gather_rows(const uint32_t *own, size_t field_h, size_t w, uint32_t *full) {
    size_t r0, r1, lo, hi;
    if (ws_tile_rows(field_h, me.rank, g->world, &r0, &r1, &lo, &hi)) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row per rank");
    if (g->is_rccl) {
      RcclApi *n = rccl();
      G_NCCL(g, n->GroupStart());
      if (me.rank != 0) {
        if (r1 > r0 && w) G_NCCL(g, n->Send(own, (r1 - r0) * w, ncclUint32, 0, g->comm, stream()));
      } else {
        for (int r = 1; r < g->world; ++r) {
          size_t a, b, l2, h2;
          (void)ws_tile_rows(field_h, r, g->world, &a, &b, &l2, &h2);
          if (b > a && w) G_NCCL(g, n->Recv(full + a * w, (b - a) * w, ncclUint32, r, g->comm, stream()));
        }
      }
      G_NCCL(g, n->GroupEnd());
      if (me.rank == 0 && r1 > r0 && w) G_HIP(g, hipMemcpyAsync(full + r0 * w, own, (r1 - r0) * w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
      G_HIP(g, hipStreamSynchronize(stream()));
      return WS_OK;
    }
    me.pub_send = own;      // (every rank's plane is complete in memory: the steps before end with a wait for their stream)
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    if (me.rank == 0) {
      for (int r = 0; r < g->world; ++r) {
        size_t a, b, l2, h2;
        (void)ws_tile_rows(field_h, r, g->world, &a, &b, &l2, &h2);
        if (b > a && w) G_HIP(g, hipMemcpyAsync(full + a * w, local(r)->pub_send, (b - a) * w * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
      }
      G_HIP(g, hipStreamSynchronize(stream()));
    }
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    return WS_OK;
  }

  // ... of a field in py x px tiles: every rank's OWNED rectangle of its plane (pitch bw, first row lo, first column clo of the
  // field) -> rank 0's whole plane (field_h x field_w).  A rectangle travels packed (me.rows); rank 0 parks what it receives in
  // me.table and places every rectangle with a 2-D copy.  Returns with rank 0's copies complete.
  int gather_rect(const uint32_t *plane, size_t bw, size_t field_h, size_t field_w, int py, int px, uint32_t *full) {
    size_t rows[4], cols[4];
    if (ws_tile_grid(field_h, field_w, me.rank, py, px, rows, cols)) return gfail(g, WS_ERR_BAD_ARG, "bad tile grid");
    const size_t oh = rows[1] - rows[0], ow = cols[1] - cols[0];
    int rc;
    if ((rc = grow(g, me.rows, std::max<size_t>(oh * ow, 1) * sizeof(uint32_t)))) return rc;
    if (oh * ow)
      G_HIP(g, hipMemcpy2DAsync(me.rows.p, ow * sizeof(uint32_t), plane + (rows[0] - rows[2]) * bw + (cols[0] - cols[2]), bw * sizeof(uint32_t),
                                ow * sizeof(uint32_t), oh, hipMemcpyDeviceToDevice, stream()));
    auto place = [&](int r, const uint32_t *packed) -> int {      // rank r's packed rectangle into the whole plane
      size_t rr[4], cc[4];
      (void)ws_tile_grid(field_h, field_w, r, py, px, rr, cc);
      const size_t h2 = rr[1] - rr[0], w2 = cc[1] - cc[0];
      if (h2 * w2)
        G_HIP(g, hipMemcpy2DAsync(full + rr[0] * field_w + cc[0], field_w * sizeof(uint32_t), packed, w2 * sizeof(uint32_t), w2 * sizeof(uint32_t), h2,
                                  hipMemcpyDeviceToDevice, stream()));
      return WS_OK;
    };
    if (g->is_rccl) {
      RcclApi *n = rccl();
      std::vector<size_t> at((size_t)g->world + 1, 0);
      if (me.rank == 0) {
        for (int r = 1; r < g->world; ++r) {
          size_t rr[4], cc[4];
          (void)ws_tile_grid(field_h, field_w, r, py, px, rr, cc);
          at[(size_t)r + 1] = at[(size_t)r] + (rr[1] - rr[0]) * (cc[1] - cc[0]);
        }
        if ((rc = grow(g, me.table, std::max<size_t>(at[(size_t)g->world], 1) * sizeof(uint32_t)))) return rc;
      }
      G_NCCL(g, n->GroupStart());
      if (me.rank != 0) {
        if (oh * ow) G_NCCL(g, n->Send(me.rows.p, oh * ow, ncclUint32, 0, g->comm, stream()));
      } else {
        for (int r = 1; r < g->world; ++r)
          if (at[(size_t)r + 1] > at[(size_t)r])
            G_NCCL(g, n->Recv((uint32_t *)me.table.p + at[(size_t)r], at[(size_t)r + 1] - at[(size_t)r], ncclUint32, r, g->comm, stream()));
      }
      G_NCCL(g, n->GroupEnd());
      if (me.rank == 0) {
        if ((rc = place(0, (const uint32_t *)me.rows.p))) return rc;
        for (int r = 1; r < g->world; ++r)
          if ((rc = place(r, (const uint32_t *)me.table.p + at[(size_t)r]))) return rc;
      }
      G_HIP(g, hipStreamSynchronize(stream()));
      return WS_OK;
    }
    G_HIP(g, hipStreamSynchronize(stream()));      // my packed rectangle is complete before rank 0 copies it
    me.pub_send = (const uint32_t *)me.rows.p;
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    if (me.rank == 0) {
      for (int r = 0; r < g->world; ++r)
        if ((rc = place(r, local(r)->pub_send))) return rc;
      G_HIP(g, hipStreamSynchronize(stream()));
    }
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    return WS_OK;
  }

  // max (or min) over the ranks of the word at me.flag[0]; stream ordered behind whatever wrote it; ONE host read
  int reduce(bool take_max, uint32_t *result) {
    if (g->is_rccl) {
      G_NCCL(g, rccl()->AllReduce(me.flag, me.flag + 1, 1, ncclUint32, take_max ? ncclMax : ncclMin, g->comm, stream()));
      G_HIP(g, hipMemcpyAsync(me.flag_host, me.flag + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
      G_HIP(g, hipStreamSynchronize(stream()));
      *result = me.flag_host[0];
      return WS_OK;
    }
    G_HIP(g, hipMemcpyAsync(me.flag_host, me.flag, sizeof(uint32_t), hipMemcpyDeviceToHost, stream()));
    G_HIP(g, hipStreamSynchronize(stream()));
    me.pub_word = me.flag_host[0];
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    uint32_t v = me.pub_word;
    for (const Rank &r : g->ranks) v = take_max ? std::max(v, r.pub_word) : std::min(v, r.pub_word);
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");      // (pub_word may be rewritten after this)
    *result = v;
    return WS_OK;
  }
  int reduce_host(uint32_t value, bool take_max, uint32_t *result) {
    me.flag_host[2] = value;
    G_HIP(g, hipMemcpyAsync(me.flag, me.flag_host + 2, sizeof(uint32_t), hipMemcpyHostToDevice, stream()));
    return reduce(take_max, result);
  }

  // `words` u32 at `send` of every rank -> table[rank * words ...] on every rank (table: world * words, on my device)
  int gather(const uint32_t *send, size_t words, uint32_t *table) {
    if (g->is_rccl) {
      G_NCCL(g, rccl()->AllGather(send, table, words, ncclUint32, g->comm, stream()));
      return WS_OK;
    }
    G_HIP(g, hipStreamSynchronize(stream()));      // my part is complete before anyone copies it
    me.pub_send = send;
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    for (const Rank &r : g->ranks)
      G_HIP(g, hipMemcpyAsync(table + (size_t)r.rank * words, r.pub_send, words * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream()));
    G_HIP(g, hipStreamSynchronize(stream()));
    if (!g->barrier.wait()) return gfail(g, WS_ERR_HIP, "another rank of the group failed");
    return WS_OK;
  }
};

// ---- the rank program: distributed.py's segment_tiled / merge_tiled -------------------------------------------------------
//
// Stamps: ws_block_begin relaxes the block to LOCAL convergence from its seed tables; then rounds of { swap halo rows;
// "did any rank receive a row that differs from the one it holds" -- compared on the device, max-reduced there (RCCL) and
// read once; copy the received rows in; ws_block_relax_halo (only the tile rows next to the halo rows start) }.
// Labels: local two-launch resolve, export the two boundary rows, ONE gather of 2 w words per rank, import.
// Merging: a union-find over all seed colours per rank, local unions, ONE gather of 4 w (colour, root) pairs, relabel.
// Lists that are not strictly increasing / widths that are not multiples of 4: the general form, decided by all ranks
// together (min-reduce of "my block can take the fast form").
int tiled_rank(ws_group *g, Rank &me, size_t field_h, size_t w, size_t n_seeds_total, const ws_tile_block &b, const ws_options *opt,
               int merging, uint32_t *rounds_out) {
  G_HIP(g, hipSetDevice(me.device));
  const int world = g->world;
  size_t r0, r1, lo, hi;
  int rc = ws_tile_rows(field_h, me.rank, world, &r0, &r1, &lo, &hi);
  if (rc) return gfail(g, rc, "a field needs at least one row per rank");
  const size_t h = hi - lo, n = h * w;
  const int up = me.rank > 0, down = me.rank < world - 1;
  uint32_t rounds = 0;
  Exchange x{g, me};
  if ((rc = grow(g, me.keys, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = grow(g, me.recv, (w ? 2 * w : 1) * sizeof(uint32_t)))) return rc;
  uint32_t *keys = (uint32_t *)me.keys.p, *recv = (uint32_t *)me.recv.p;
  hipStream_t s = me.ctx->stream;

  // can this block take the fast form?  (the order of the list is checked by ws_block_begin, on the device)
  uint32_t fast = b.d_colours == nullptr && (w & 3) == 0 && h >= 2 && n < 0x80000000ull && w > 0 ? 1u : 0u;
  if (fast) {
    rc = ws_block_begin(me.ctx, b.d_img, h, w, w, opt->max_water_level, b.d_seeds_rc, b.n_seeds, b.first_colour, keys);
    if (rc == WS_ERR_UNSUPPORTED) fast = 0;
    else if (rc != WS_OK) return gfail(g, rc, std::string("rank ") + std::to_string(me.rank) + ": ws_block_begin: " + ws_last_error(me.ctx));
  }
  uint32_t all_fast = 0;
  if ((rc = x.reduce_host(fast, false, &all_fast))) return rc;
  ++rounds;
  if (all_fast) {
    for (;;) {
      if ((rc = x.swap(keys, h, w))) return rc;
      ++rounds;
      G_HIP(g, hipMemsetAsync(me.flag, 0, sizeof(uint32_t), s));
      if (up) G_HIP(g, block_rows_differ(s, recv, keys, w, me.flag));
      if (down) G_HIP(g, block_rows_differ(s, recv + w, keys + (h - 1) * w, w, me.flag));
      uint32_t any = 0;
      if ((rc = x.reduce(true, &any))) return rc;
      if (!any) break;      // every halo row already equals its neighbour's boundary row
      if (up) G_HIP(g, hipMemcpyAsync(keys, recv, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
      if (down) G_HIP(g, hipMemcpyAsync(keys + (h - 1) * w, recv + w, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
      G_WS(g, me, ws_block_relax_halo(me.ctx, b.d_img, h, w, w, opt->max_water_level, up, down, keys));
    }
    G_WS(g, me, ws_block_resolve_local(me.ctx, keys, b.d_labels, h, w, up, down));
    if ((rc = grow(g, me.rows, 2 * w * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.table, (size_t)world * 2 * w * sizeof(uint32_t)))) return rc;
    G_WS(g, me, ws_block_export_boundary(me.ctx, b.d_labels, h, w, up, down, (size_t)me.rank, (uint32_t *)me.rows.p));
    if ((rc = x.gather((const uint32_t *)me.rows.p, 2 * w, (uint32_t *)me.table.p))) return rc;
    ++rounds;
    G_WS(g, me, ws_block_import_boundary(me.ctx, (const uint32_t *)me.table.p, (size_t)world, (size_t)me.rank, b.d_labels, h, w, up, down));
  } else {
    // general form: painted seeds, relaxation rounds and label rounds, one halo swap and one flag each
    const uint32_t *colours = b.d_colours;
    if (!colours && b.n_seeds) {
      if ((rc = grow(g, me.colours, b.n_seeds * sizeof(uint32_t)))) return rc;
      G_HIP(g, block_iota(s, (uint32_t *)me.colours.p, b.n_seeds, b.first_colour));
      colours = (const uint32_t *)me.colours.p;
    }
    G_WS(g, me, ws_block_init(me.ctx, h, w, b.d_seeds_rc, colours, b.n_seeds, keys, b.d_labels));
    for (int phase = 0; phase < 2; ++phase) {
      uint32_t *plane = phase == 0 ? keys : b.d_labels;
      for (;;) {
        int changed = 0;
        if (phase == 0) G_WS(g, me, ws_block_relax(me.ctx, b.d_img, h, w, w, opt->max_water_level, keys, &changed));
        else G_WS(g, me, ws_block_resolve(me.ctx, keys, b.d_labels, h, w, &changed));
        if ((rc = x.swap(plane, h, w))) return rc;
        if (up) G_HIP(g, hipMemcpyAsync(plane, recv, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        if (down) G_HIP(g, hipMemcpyAsync(plane + (h - 1) * w, recv + w, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
        uint32_t any = 0;
        if ((rc = x.reduce_host(changed ? 1u : 0u, true, &any))) return rc;
        ++rounds;
        if (!any) break;
      }
    }
  }
  if (merging) {
    const size_t n_pairs = 4 * w;
    if ((rc = grow(g, me.parent, (n_seeds_total + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.rows, n_pairs * 2 * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.table, (size_t)world * n_pairs * 2 * sizeof(uint32_t)))) return rc;
    uint32_t *parent = (uint32_t *)me.parent.p;
    G_WS(g, me, ws_block_merge_local(me.ctx, b.d_labels, h, w, lo, field_h, n_seeds_total, parent));
    G_WS(g, me, ws_block_merge_export(me.ctx, b.d_labels, h, w, parent, (uint32_t *)me.rows.p));
    if ((rc = x.gather((const uint32_t *)me.rows.p, n_pairs * 2, (uint32_t *)me.table.p))) return rc;
    ++rounds;
    G_WS(g, me, ws_block_merge_import(me.ctx, (const uint32_t *)me.table.p, (size_t)world * n_pairs, parent));
    G_WS(g, me, ws_block_merge_relabel(me.ctx, b.d_labels, n, parent, n_seeds_total, b.d_labels));      // elementwise: in place
  }
  G_HIP(g, hipStreamSynchronize(s));
  if (rounds_out) *rounds_out = rounds;
  return WS_OK;
}

// A field cut into py x px tiles (BASELINE config 5's "2-D tiles"; rank = ty * px + tx): every tile is a plane with a halo
// ring of one pixel wherever it has a neighbour -- exactly the pixels the flood never writes, rows AND columns
// (lib.rs:220-222), so the block steps of the general form run on it unchanged: painted seeds with their global colours,
// rounds of { relax to local convergence; swap halo rows and columns; "did any rank change anything" }, then the same
// rounds for the labels (one hop per sweep).  Halo traffic per round: 2 (w + h) words a tile.
int tiled2d_rank(ws_group *g, Rank &me, size_t field_h, size_t field_w, int py, int px, size_t n_seeds_total, const ws_tile_block2d &b,
                 const ws_options *opt, int merging, uint32_t *rounds_out) {
  G_HIP(g, hipSetDevice(me.device));
  const int ty = me.rank / px, tx = me.rank % px;
  size_t r0, r1, lo, hi, c0, c1, clo, chi;
  if (ws_tile_rows(field_h, ty, py, &r0, &r1, &lo, &hi) || ws_tile_rows(field_w, tx, px, &c0, &c1, &clo, &chi))
    return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row and one column per tile");
  const size_t h = hi - lo, w = chi - clo, n = h * w;
  const int nb[4] = {ty > 0 ? me.rank - px : -1, ty < py - 1 ? me.rank + px : -1, tx > 0 ? me.rank - 1 : -1, tx < px - 1 ? me.rank + 1 : -1};
  uint32_t rounds = 0;
  Exchange x{g, me};
  int rc;
  if ((rc = grow(g, me.keys, (n ? n : 1) * sizeof(uint32_t)))) return rc;
  if ((rc = grow(g, me.recv, (2 * w + 2 * h + 1) * sizeof(uint32_t)))) return rc;
  if ((rc = grow(g, me.cols, (2 * h + 1) * sizeof(uint32_t)))) return rc;
  uint32_t *keys = (uint32_t *)me.keys.p, *recv = (uint32_t *)me.recv.p;
  hipStream_t s = me.ctx->stream;
  G_WS(g, me, ws_block_init(me.ctx, h, w, b.d_seeds_rc, b.d_colours, b.n_seeds, keys, b.d_labels));
  auto copy_ring_in = [&](uint32_t *plane) -> int {
    if (nb[0] >= 0) G_HIP(g, hipMemcpyAsync(plane, recv, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    if (nb[1] >= 0) G_HIP(g, hipMemcpyAsync(plane + (h - 1) * w, recv + w, w * sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    // (columns after rows: a halo column's first / last cell is a corner of the ring, read by nobody)
    G_HIP(g, block_unpack_cols(s, plane, h, w, nb[2] >= 0 ? recv + 2 * w : nullptr, nb[3] >= 0 ? recv + 2 * w + h : nullptr));
    return WS_OK;
  };
  // stamps: relax to local convergence, swap the ring, until no rank changed anything
  for (;;) {
    int changed = 0;
    G_WS(g, me, ws_block_relax(me.ctx, b.d_img, h, w, b.img_stride, opt->max_water_level, keys, &changed));
    if ((rc = x.swap2d(keys, h, w, nb))) return rc;
    if ((rc = copy_ring_in(keys))) return rc;
    uint32_t any = 0;
    if ((rc = x.reduce_host(changed ? 1u : 0u, true, &any))) return rc;
    ++rounds;
    if (!any) break;
  }
  // labels: the whole tile in two launches from its seeds and from what the ring says so far (ws_block_resolve_ring), swap
  // the ring, until no rank receives a ring that differs from the one it holds: a round carries labels across one tile
  // boundary.  (The iterative resolve -- one hop per launch -- took 32 launches per rank where this takes 3 rounds of 2.)
  if (n >= 0x80000000ull) {
    for (;;) {      // planes of 2^31 pixels and more: the iterative form
      int changed = 0;
      G_WS(g, me, ws_block_resolve(me.ctx, keys, b.d_labels, h, w, &changed));
      if ((rc = x.swap2d(b.d_labels, h, w, nb))) return rc;
      if ((rc = copy_ring_in(b.d_labels))) return rc;
      uint32_t any = 0;
      if ((rc = x.reduce_host(changed ? 1u : 0u, true, &any))) return rc;
      ++rounds;
      if (!any) break;
    }
  } else {
    if ((rc = grow(g, me.rows, (2 * h + 1) * sizeof(uint32_t)))) return rc;
    uint32_t *held = (uint32_t *)me.rows.p;      // the halo columns I hold, packed like the received ones
    for (;;) {
      G_WS(g, me, ws_block_resolve_ring(me.ctx, keys, b.d_labels, h, w));
      if ((rc = x.swap2d(b.d_labels, h, w, nb))) return rc;
      G_HIP(g, hipMemsetAsync(me.flag, 0, sizeof(uint32_t), s));
      // (the four corner cells of the ring are left out: the row copy and the column copy both write them, with different
      // ranks' values, and nobody's stencil reads them)
      if (nb[0] >= 0 && w > 2) G_HIP(g, block_rows_differ(s, recv + 1, b.d_labels + 1, w - 2, me.flag));
      if (nb[1] >= 0 && w > 2) G_HIP(g, block_rows_differ(s, recv + w + 1, b.d_labels + (h - 1) * w + 1, w - 2, me.flag));
      if (nb[2] >= 0 || nb[3] >= 0) {
        G_HIP(g, block_pack_cols(s, b.d_labels, h, w, 0, w - 1, held));
        if (nb[2] >= 0 && h > 2) G_HIP(g, block_rows_differ(s, recv + 2 * w + 1, held + 1, h - 2, me.flag));
        if (nb[3] >= 0 && h > 2) G_HIP(g, block_rows_differ(s, recv + 2 * w + h + 1, held + h + 1, h - 2, me.flag));
      }
      uint32_t any = 0;
      if ((rc = x.reduce(true, &any))) return rc;
      ++rounds;
      if (!any) break;
      if ((rc = copy_ring_in(b.d_labels))) return rc;
    }
  }
  if (merging) {
    // as the row blocks' (ws_block_merge_*): a union-find over all seed colours per rank, the touching colours of the tile
    // joined under find_merge's rule (one pixel of a pair interior IN THE WHOLE FIELD), then ONE gather of (colour, local
    // root) pairs of the tile's two outermost rows and columns on every side, every pair joined, the relabel
    const size_t max_h = (field_h + (size_t)py - 1) / (size_t)py + 2, max_w = (field_w + (size_t)px - 1) / (size_t)px + 2;
    const size_t n_pairs = 4 * max_w + 4 * max_h;      // the same for every rank: a tile's own 4 w + 4 h, then (0, 0)
    const int world = g->world;
    if ((rc = grow(g, me.parent, (n_seeds_total + 1) * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.rows, n_pairs * 2 * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.table, (size_t)world * n_pairs * 2 * sizeof(uint32_t)))) return rc;
    if ((rc = ensure(me.ctx, me.ctx->uf_size, (n_seeds_total + 1) * sizeof(uint32_t)))) return gfail(g, rc, "out of memory");
    uint32_t *parent = (uint32_t *)me.parent.p;
    G_HIP(g, uf_init(s, parent, (uint32_t *)me.ctx->uf_size.p, n_seeds_total + 1));
    G_HIP(g, block_union_pixels(s, b.d_labels, (int)h, (int)w, (int)lo, (int)field_h, parent, (int)clo, (int)field_w));
    G_HIP(g, block_colour_roots2d(s, b.d_labels, (int)h, (int)w, parent, (uint2 *)me.rows.p, n_pairs));
    if ((rc = x.gather((const uint32_t *)me.rows.p, n_pairs * 2, (uint32_t *)me.table.p))) return rc;
    ++rounds;
    G_HIP(g, union_edges(s, (const uint2 *)me.table.p, (size_t)world * n_pairs, parent, nullptr, nullptr));
    G_HIP(g, relabel_final_u32(s, b.d_labels, parent, n_seeds_total + 1, b.d_labels, n));      // elementwise: in place
  }
  G_HIP(g, hipStreamSynchronize(s));
  if (rounds_out) *rounds_out = rounds;
  return WS_OK;
}

// world == 1: the block is the whole field -- the ordinary single-device transform
int single_rank(ws_group *g, Rank &me, size_t field_h, size_t w, const ws_tile_block &b, const ws_options *opt, int merging) {
  if (b.d_colours) return gfail(g, WS_ERR_UNSUPPORTED, "a group of one rank takes the caller's list as it is: d_colours must be NULL");
  if (merging) G_WS(g, me, ws_merge_device(me.ctx, b.d_img, field_h, w, w, b.d_seeds_rc, b.n_seeds, opt, b.d_labels));
  else G_WS(g, me, ws_segment_device(me.ctx, b.d_img, field_h, w, w, b.d_seeds_rc, b.n_seeds, opt, b.d_labels));
  return WS_OK;
}

// runs f(rank) for every local rank: side by side on host threads for a local group of several ranks
template <class F>
int for_local_ranks(ws_group *g, F f) {
  const size_t nl = g->ranks.size();
  g->barrier.reset((int)nl);
  { std::lock_guard<std::mutex> l(g->err_m); g->err.clear(); }
  if (nl == 1) return f(g->ranks[0]);
  std::vector<int> rcs(nl, WS_OK);
  std::vector<std::thread> th;
  for (size_t i = 0; i < nl; ++i)
    th.emplace_back([&, i] {
      rcs[i] = f(g->ranks[i]);
      if (rcs[i] != WS_OK) g->barrier.abort();      // nobody waits for a rank that has left
    });
  for (std::thread &t : th) t.join();
  // the rank that failed first in its own right, not the ones that were told "another rank failed"
  int first = WS_OK;
  for (int rc : rcs) if (rc != WS_OK && (first == WS_OK || first == WS_ERR_HIP)) first = rc;
  return first;
}

int check_group_call(ws_group *g, const ws_options *opt) {
  if (!g) return WS_ERR_BAD_ARG;
  if (!opt) return gfail(g, WS_ERR_BAD_ARG, "options pointer is null");
  const int v = ws_options_validate(opt);
  if (v != WS_OK) return gfail(g, v, ws_strerror(v));
  if (pick_engine(opt) == WS_ENGINE_SWEEP && g->world > 1) return gfail(g, WS_ERR_UNSUPPORTED, "the sweep engine has no tiled form");
  return WS_OK;
}

}  // namespace

// ============================================================================ C ABI ====

extern "C" {

int ws_tile_rows(size_t h, int rank, int world, size_t *r0, size_t *r1, size_t *lo, size_t *hi) {
  if (world < 1 || rank < 0 || rank >= world || !r0 || !r1 || !lo || !hi) return WS_ERR_BAD_ARG;
  if (h < (size_t)world) return WS_ERR_BAD_ARG;      // a rank without rows would hand a halo row on as if it were its own
  const size_t base = h / (size_t)world, extra = h % (size_t)world, r = (size_t)rank;
  *r0 = r * base + std::min(r, extra);
  *r1 = *r0 + base + (r < extra ? 1 : 0);
  *lo = rank > 0 ? *r0 - 1 : *r0;
  *hi = rank < world - 1 ? *r1 + 1 : *r1;
  return WS_OK;
}

static int group_alloc_rank(ws_group *g, Rank &r) {
  G_HIP(g, hipSetDevice(r.device));
  const int rc = ws_ctx_create(r.device, &r.ctx);
  if (rc != WS_OK) return gfail(g, rc, "ws_ctx_create failed");
  G_HIP(g, hipMalloc((void **)&r.flag, 4 * sizeof(uint32_t)));
  G_HIP(g, hipMemset(r.flag, 0, 4 * sizeof(uint32_t)));
  G_HIP(g, hipHostMalloc((void **)&r.flag_host, 4 * sizeof(uint32_t), hipHostMallocDefault));
  return WS_OK;
}

int ws_group_create_local(int n_ranks, const int *devices, ws_group **out) {
  if (!out || n_ranks < 1 || n_ranks > 1024) return WS_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return WS_ERR_NO_DEVICE;
  for (int r = 0; r < n_ranks; ++r)
    if (devices && (devices[r] < 0 || devices[r] >= count)) return WS_ERR_BAD_ARG;
  ws_group *g = new (std::nothrow) ws_group();
  if (!g) return WS_ERR_OOM;
  g->world = n_ranks;
  g->ranks.resize((size_t)n_ranks);
  int rc = WS_OK;
  for (int r = 0; r < n_ranks && rc == WS_OK; ++r) {
    g->ranks[(size_t)r].rank = r;
    g->ranks[(size_t)r].device = devices ? devices[r] : 0;
    rc = group_alloc_rank(g, g->ranks[(size_t)r]);
  }
  // neighbours on different devices copy each other's rows directly
  for (int r = 0; r + 1 < n_ranks && rc == WS_OK; ++r) {
    const int a = g->ranks[(size_t)r].device, b = g->ranks[(size_t)r + 1].device;
    if (a == b) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) { (void)hipSetDevice(a); (void)hipDeviceEnablePeerAccess(b, 0); }
    if (hipDeviceCanAccessPeer(&can, b, a) == hipSuccess && can) { (void)hipSetDevice(b); (void)hipDeviceEnablePeerAccess(a, 0); }
    (void)hipGetLastError();      // (already enabled: fine; not possible: the copies are staged by the runtime)
  }
  if (rc != WS_OK) { ws_group_destroy(g); return rc; }
  *out = g;
  return WS_OK;
}

int ws_group_rccl_unique_id(void *id) {
  if (!id) return WS_ERR_BAD_ARG;
  static_assert(sizeof(ncclUniqueId) == WS_RCCL_ID_BYTES, "the id travels as WS_RCCL_ID_BYTES opaque bytes");
  RcclApi *n = rccl();
  if (!n->lib) return WS_ERR_RCCL;
  ncclUniqueId uid;
  if (n->GetUniqueId(&uid) != ncclSuccess) return WS_ERR_RCCL;
  std::memcpy(id, &uid, sizeof uid);
  return WS_OK;
}

int ws_group_create_rccl(int device, int rank, int world, const void *id, ws_group **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return WS_ERR_BAD_ARG;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return WS_ERR_NO_DEVICE;
  if (device < 0 || device >= count) return WS_ERR_BAD_ARG;
  RcclApi *n = rccl();
  if (!n->lib) return WS_ERR_RCCL;
  ws_group *g = new (std::nothrow) ws_group();
  if (!g) return WS_ERR_OOM;
  g->is_rccl = true;
  g->world = world;
  g->first_local = rank;
  g->ranks.resize(1);
  g->ranks[0].rank = rank;
  g->ranks[0].device = device;
  int rc = group_alloc_rank(g, g->ranks[0]);
  if (rc == WS_OK) {
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof uid);
    if (hipSetDevice(device) != hipSuccess || n->CommInitRank(&g->comm, world, uid, rank) != ncclSuccess) { g->comm = nullptr; rc = WS_ERR_RCCL; }
  }
  if (rc != WS_OK) { ws_group_destroy(g); return rc; }
  *out = g;
  return WS_OK;
}

void ws_group_destroy(ws_group *g) {
  if (!g) return;
  if (g->comm) (void)rccl()->CommDestroy(g->comm);
  for (Rank &r : g->ranks) {
    (void)hipSetDevice(r.device);
    if (r.ctx) { (void)ws_ctx_synchronize(r.ctx); }
    for (Grow *b : {&r.keys, &r.labels, &r.recv, &r.rows, &r.table, &r.parent, &r.img, &r.seeds, &r.colours, &r.out64})
      if (b->p) (void)hipFree(b->p);
    if (r.flag) (void)hipFree(r.flag);
    if (r.flag_host) (void)hipHostFree(r.flag_host);
    if (r.ctx) ws_ctx_destroy(r.ctx);
  }
  delete g;
}

int ws_group_info(const ws_group *g, int *world, int *n_local, int *first_local) {
  if (!g) return WS_ERR_BAD_ARG;
  if (world) *world = g->world;
  if (n_local) *n_local = (int)g->ranks.size();
  if (first_local) *first_local = g->first_local;
  return WS_OK;
}

const char *ws_group_last_error(const ws_group *g) {
  if (!g) return rccl()->lib ? "null group" : rccl()->why.c_str();
  return g->err.c_str();
}

int ws_group_selftest(ws_group *g) {
  if (!g) return WS_ERR_BAD_ARG;
  const size_t w = 1024, h = 4;
  return for_local_ranks(g, [&](Rank &me) -> int {
    G_HIP(g, hipSetDevice(me.device));
    int rc;
    const int world = g->world, up = me.rank > 0, down = me.rank < world - 1;
    if ((rc = grow(g, me.keys, h * w * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.recv, 2 * w * sizeof(uint32_t)))) return rc;
    if ((rc = grow(g, me.table, (size_t)world * w * sizeof(uint32_t)))) return rc;
    std::vector<uint32_t> host(h * w), got(std::max<size_t>(2 * w, (size_t)world * w));
    for (size_t i = 0; i < h * w; ++i) host[i] = (uint32_t)me.rank * 100000u + (uint32_t)i;      // row r of rank k: k * 100000 + r * w ...
    hipStream_t s = me.ctx->stream;
    G_HIP(g, hipMemcpyAsync(me.keys.p, host.data(), h * w * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    G_HIP(g, hipStreamSynchronize(s));
    Exchange x{g, me};
    if ((rc = x.swap((const uint32_t *)me.keys.p, h, w))) return rc;
    G_HIP(g, hipMemcpyAsync(got.data(), me.recv.p, 2 * w * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    G_HIP(g, hipStreamSynchronize(s));
    // the upper neighbour's last OWNED row (its row h - 2 if it has a lower neighbour -- it has: me), the lower one's first owned row
    for (size_t i = 0; i < w; ++i) {
      if (up && got[i] != (uint32_t)(me.rank - 1) * 100000u + (uint32_t)((h - 2) * w + i)) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: halo swap (from above) delivered wrong words");
      if (down && got[w + i] != (uint32_t)(me.rank + 1) * 100000u + (uint32_t)(w + i)) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: halo swap (from below) delivered wrong words");
    }
    uint32_t v = 0;
    if ((rc = x.reduce_host((uint32_t)me.rank + 7u, true, &v))) return rc;
    if (v != (uint32_t)world + 6u) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: max-reduce returned a wrong word");
    if ((rc = x.reduce_host((uint32_t)me.rank + 7u, false, &v))) return rc;
    if (v != 7u) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: min-reduce returned a wrong word");
    if ((rc = x.gather((const uint32_t *)me.keys.p + w, w, (uint32_t *)me.table.p))) return rc;
    G_HIP(g, hipMemcpyAsync(got.data(), me.table.p, (size_t)world * w * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    G_HIP(g, hipStreamSynchronize(s));
    for (int k = 0; k < world; ++k)
      for (size_t i = 0; i < w; ++i)
        if (got[(size_t)k * w + i] != (uint32_t)k * 100000u + (uint32_t)(w + i)) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: all-gather delivered wrong words");
    // the gather of owned rows on rank 0 (transform_to_list of a tiled field): a field of 2 * world + 1 rows, every rank's owned
    // rows hold rank * 100000 + field row * w + column
    {
      const size_t fh = 2 * (size_t)world + 1;
      size_t r0, r1, lo, hi;
      if (ws_tile_rows(fh, me.rank, world, &r0, &r1, &lo, &hi)) return gfail(g, WS_ERR_BAD_ARG, "selftest: tile rows");
      std::vector<uint32_t> own((r1 - r0) * w), full(fh * w);
      for (size_t r = r0; r < r1; ++r)
        for (size_t i = 0; i < w; ++i) own[(r - r0) * w + i] = (uint32_t)me.rank * 100000u + (uint32_t)(r * w + i);
      if ((rc = grow(g, me.rows, std::max<size_t>(own.size(), 1) * sizeof(uint32_t)))) return rc;
      if (me.rank == 0 && (rc = grow(g, me.full_keys, fh * w * sizeof(uint32_t)))) return rc;
      G_HIP(g, hipMemcpyAsync(me.rows.p, own.data(), own.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      G_HIP(g, hipStreamSynchronize(s));
      if ((rc = x.gather_rows((const uint32_t *)me.rows.p, fh, w, (uint32_t *)me.full_keys.p))) return rc;
      if (me.rank == 0) {
        G_HIP(g, hipMemcpyAsync(full.data(), me.full_keys.p, fh * w * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        G_HIP(g, hipStreamSynchronize(s));
        for (int k = 0; k < world; ++k) {
          size_t a, b, l2, h2;
          (void)ws_tile_rows(fh, k, world, &a, &b, &l2, &h2);
          for (size_t r = a; r < b; ++r)
            for (size_t i = 0; i < w; ++i)
              if (full[r * w + i] != (uint32_t)k * 100000u + (uint32_t)(r * w + i)) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: the gather of owned rows on rank 0 delivered wrong words");
        }
      }
    }
    // ... and of owned rectangles (transform_to_list of a field in tiles): a field of 3 x (2 * world + 1) words cut into 1 x world
    // tiles, every word rank * 100000 + its index in the field
    {
      const size_t fh = 3, fw = 2 * (size_t)world + 1;
      size_t rows[4], cols[4];
      if (ws_tile_grid(fh, fw, me.rank, 1, world, rows, cols)) return gfail(g, WS_ERR_BAD_ARG, "selftest: tile grid");
      const size_t bh = rows[3] - rows[2], bw = cols[3] - cols[2];
      std::vector<uint32_t> plane(bh * bw, 0xDEADu), full(fh * fw);
      for (size_t r = rows[0]; r < rows[1]; ++r)
        for (size_t q = cols[0]; q < cols[1]; ++q) plane[(r - rows[2]) * bw + (q - cols[2])] = (uint32_t)me.rank * 100000u + (uint32_t)(r * fw + q);
      if ((rc = grow(g, me.keys, std::max<size_t>(plane.size(), h * w) * sizeof(uint32_t)))) return rc;
      if (me.rank == 0 && (rc = grow(g, me.full_keys, std::max<size_t>(fh * fw, (2 * (size_t)world + 1) * w) * sizeof(uint32_t)))) return rc;
      G_HIP(g, hipMemcpyAsync(me.keys.p, plane.data(), plane.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
      G_HIP(g, hipStreamSynchronize(s));
      if ((rc = x.gather_rect((const uint32_t *)me.keys.p, bw, fh, fw, 1, world, (uint32_t *)me.full_keys.p))) return rc;
      if (me.rank == 0) {
        G_HIP(g, hipMemcpyAsync(full.data(), me.full_keys.p, fh * fw * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        G_HIP(g, hipStreamSynchronize(s));
        for (int k = 0; k < world; ++k) {
          size_t rr[4], cc[4];
          (void)ws_tile_grid(fh, fw, k, 1, world, rr, cc);
          for (size_t r = rr[0]; r < rr[1]; ++r)
            for (size_t q = cc[0]; q < cc[1]; ++q)
              if (full[r * fw + q] != (uint32_t)k * 100000u + (uint32_t)(r * fw + q)) return gfail(g, g->is_rccl ? WS_ERR_RCCL : WS_ERR_HIP, "selftest: the gather of owned rectangles on rank 0 delivered wrong words");
        }
      }
    }
    return WS_OK;
  });
}

int ws_segment_tiled_device(ws_group *g, size_t field_h, size_t w, size_t n_seeds_total, const ws_tile_block *blocks, const ws_options *opt,
                            int merging, uint32_t *exchange_rounds) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (!blocks) return gfail(g, WS_ERR_BAD_ARG, "blocks pointer is null");
  if (opt->edge_correction) return gfail(g, WS_ERR_UNSUPPORTED, "ws_segment_tiled_device takes the field as it is: pad it first (ws_segment_tiled does)");
  if (n_seeds_total >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "colours must stay below 2^31");
  if (field_h < (size_t)g->world) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row per rank");
  for (size_t i = 0; i < g->ranks.size(); ++i) {
    const ws_tile_block &b = blocks[i];
    if (b.reserved != 0 || (field_h * w && (!b.d_img || !b.d_labels)) || (b.n_seeds && !b.d_seeds_rc))
      return gfail(g, WS_ERR_BAD_ARG, "bad block descriptor");
  }
  if (exchange_rounds) *exchange_rounds = 0;
  if (g->world == 1) return for_local_ranks(g, [&](Rank &me) { return single_rank(g, me, field_h, w, blocks[0], opt, merging); });
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) {
    const size_t i = (size_t)(me.rank - g->first_local);
    return tiled_rank(g, me, field_h, w, n_seeds_total, blocks[i], opt, merging, &rounds[i]);
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

// transform_to_list (lib.rs:1551-1561 merging, 1837-1847 segmenting) of a field cut into row blocks: the flood -- the part that
// scales with the pixels and needs the halo exchange -- runs on all ranks (tiled_rank), then every rank sends the arrival stamps
// and segment labels of its owned rows to rank 0, whose context writes the lake records of all levels from the whole plane
// (ws_lists_from_arrival_device: 8 B per pixel gathered, 20 B per pixel of workspace -- a 32768^2 field is 28 GiB of one
// device's 288).  The lists are global objects (a lake spans blocks, its area is a sum over them); cutting the union-find
// itself across ranks would trade this one gather for an exchange per level.
int ws_transform_to_list_tiled_device(ws_group *g, size_t field_h, size_t w, size_t n_seeds_total, const ws_tile_block *blocks,
                                      const ws_options *opt, int merging, ws_lake *d_lakes, size_t cap, size_t *n_lakes,
                                      uint64_t *offsets, uint64_t *uncoloured, uint32_t *exchange_rounds) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (!blocks || !n_lakes || !offsets || !uncoloured) return gfail(g, WS_ERR_BAD_ARG, "null pointer");
  if (opt->edge_correction) return gfail(g, WS_ERR_UNSUPPORTED, "the tiled transforms take the field as it is: pad it first");
  if (n_seeds_total >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "colours must stay below 2^31");
  if (field_h < (size_t)g->world) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row per rank");
  const bool root_here = g->first_local == 0;
  if (root_here && !d_lakes && cap) return gfail(g, WS_ERR_BAD_ARG, "d_lakes is null");
  for (size_t i = 0; i < g->ranks.size(); ++i) {
    const ws_tile_block &b = blocks[i];
    if (b.reserved != 0 || (field_h * w && (!b.d_img || !b.d_labels)) || (b.n_seeds && !b.d_seeds_rc))
      return gfail(g, WS_ERR_BAD_ARG, "bad block descriptor");
  }
  *n_lakes = 0;
  if (exchange_rounds) *exchange_rounds = 0;
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    const size_t i = (size_t)(me.rank - g->first_local);
    const ws_tile_block &b = blocks[i];
    int rc2;
    const uint32_t *keys, *labels;      // the rank's block: rows [lo, hi) of the field
    size_t r0, r1, lo, hi;
    if ((rc2 = ws_tile_rows(field_h, me.rank, g->world, &r0, &r1, &lo, &hi))) return gfail(g, rc2, "a field needs at least one row per rank");
    if (g->world == 1) {
      if ((rc2 = single_rank(g, me, field_h, w, b, opt, 0))) return rc2;
      const uint32_t *k = nullptr; size_t kh = 0, kw = 0;
      G_WS(g, me, ws_last_arrival_device(me.ctx, &k, &kh, &kw));
      return gfail_if(g, me, ws_lists_from_arrival_device(me.ctx, merging, k, b.d_labels, field_h, w, n_seeds_total, opt, d_lakes, cap, n_lakes, offsets, uncoloured));
    }
    if ((rc2 = tiled_rank(g, me, field_h, w, n_seeds_total, b, opt, 0, &rounds[i]))) return rc2;
    keys = (const uint32_t *)me.keys.p;
    labels = b.d_labels;
    Exchange x{g, me};
    const size_t n = field_h * w;
    if (me.rank == 0) {
      if ((rc2 = grow(g, me.full_keys, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
      if ((rc2 = grow(g, me.full_labels, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
    }
    G_HIP(g, hipStreamSynchronize(me.ctx->stream));
    if ((rc2 = x.gather_rows(keys + (r0 - lo) * w, field_h, w, (uint32_t *)me.full_keys.p))) return rc2;
    if ((rc2 = x.gather_rows(labels + (r0 - lo) * w, field_h, w, (uint32_t *)me.full_labels.p))) return rc2;
    if (me.rank != 0) return WS_OK;
    return gfail_if(g, me, ws_lists_from_arrival_device(me.ctx, merging, (const uint32_t *)me.full_keys.p, (const uint32_t *)me.full_labels.p, field_h, w,
                                                        n_seeds_total, opt, d_lakes, cap, n_lakes, offsets, uncoloured));
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

int ws_tile_grid(size_t h, size_t w, int rank, int py, int px, size_t *rows, size_t *cols) {
  if (py < 1 || px < 1 || rank < 0 || rank >= py * px || !rows || !cols) return WS_ERR_BAD_ARG;
  int rc = ws_tile_rows(h, rank / px, py, rows, rows + 1, rows + 2, rows + 3);
  if (rc == WS_OK) rc = ws_tile_rows(w, rank % px, px, cols, cols + 1, cols + 2, cols + 3);
  return rc;
}

int ws_segment_tiled2d_device(ws_group *g, size_t field_h, size_t field_w, int py, int px, size_t n_seeds_total, const ws_tile_block2d *blocks,
                              const ws_options *opt, int merging, uint32_t *exchange_rounds) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (!blocks) return gfail(g, WS_ERR_BAD_ARG, "blocks pointer is null");
  if (py < 1 || px < 1 || py * px != g->world) return gfail(g, WS_ERR_BAD_ARG, "py * px must be the group's number of ranks");
  if (opt->edge_correction) return gfail(g, WS_ERR_UNSUPPORTED, "ws_segment_tiled2d_device takes the field as it is: pad it first");
  if (field_h < (size_t)py || field_w < (size_t)px) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row and one column per tile");
  if (n_seeds_total >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "colours must stay below 2^31");
  for (size_t i = 0; i < g->ranks.size(); ++i) {
    const ws_tile_block2d &b = blocks[i];
    size_t rws[4], cls[4];
    if (ws_tile_grid(field_h, field_w, g->first_local + (int)i, py, px, rws, cls)) return gfail(g, WS_ERR_BAD_ARG, "bad tile grid");
    if (!b.d_img || !b.d_labels || b.img_stride < cls[3] - cls[2] || (b.n_seeds && (!b.d_seeds_rc || !b.d_colours)))
      return gfail(g, WS_ERR_BAD_ARG, "bad block descriptor");
  }
  if (exchange_rounds) *exchange_rounds = 0;
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) {
    const size_t i = (size_t)(me.rank - g->first_local);
    return tiled2d_rank(g, me, field_h, field_w, py, px, n_seeds_total, blocks[i], opt, merging, &rounds[i]);
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

// transform_to_list of a field in py x px tiles: as ws_transform_to_list_tiled_device, the owned RECTANGLES gathered on rank 0
int ws_transform_to_list_tiled2d_device(ws_group *g, size_t field_h, size_t field_w, int py, int px, size_t n_seeds_total,
                                        const ws_tile_block2d *blocks, const ws_options *opt, int merging, ws_lake *d_lakes, size_t cap,
                                        size_t *n_lakes, uint64_t *offsets, uint64_t *uncoloured, uint32_t *exchange_rounds) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (!blocks || !n_lakes || !offsets || !uncoloured) return gfail(g, WS_ERR_BAD_ARG, "null pointer");
  if (py < 1 || px < 1 || py * px != g->world) return gfail(g, WS_ERR_BAD_ARG, "py * px must be the group's number of ranks");
  if (opt->edge_correction) return gfail(g, WS_ERR_UNSUPPORTED, "the tiled transforms take the field as it is: pad it first");
  if (field_h < (size_t)py || field_w < (size_t)px) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row and one column per tile");
  if (n_seeds_total >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "colours must stay below 2^31");
  if (g->first_local == 0 && !d_lakes && cap) return gfail(g, WS_ERR_BAD_ARG, "d_lakes is null");
  for (size_t i = 0; i < g->ranks.size(); ++i) {
    const ws_tile_block2d &b = blocks[i];
    size_t rws[4], cls[4];
    if (ws_tile_grid(field_h, field_w, g->first_local + (int)i, py, px, rws, cls)) return gfail(g, WS_ERR_BAD_ARG, "bad tile grid");
    if (!b.d_img || !b.d_labels || b.img_stride < cls[3] - cls[2] || (b.n_seeds && (!b.d_seeds_rc || !b.d_colours)))
      return gfail(g, WS_ERR_BAD_ARG, "bad block descriptor");
  }
  *n_lakes = 0;
  if (exchange_rounds) *exchange_rounds = 0;
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    const size_t i = (size_t)(me.rank - g->first_local);
    int rc2;
    if ((rc2 = tiled2d_rank(g, me, field_h, field_w, py, px, n_seeds_total, blocks[i], opt, 0, &rounds[i]))) return rc2;
    size_t rows[4], cols[4];
    (void)ws_tile_grid(field_h, field_w, me.rank, py, px, rows, cols);
    const size_t bw = cols[3] - cols[2], n = field_h * field_w;
    if (me.rank == 0) {
      if ((rc2 = grow(g, me.full_keys, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
      if ((rc2 = grow(g, me.full_labels, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
    }
    Exchange x{g, me};
    G_HIP(g, hipStreamSynchronize(me.ctx->stream));
    if ((rc2 = x.gather_rect((const uint32_t *)me.keys.p, bw, field_h, field_w, py, px, (uint32_t *)me.full_keys.p))) return rc2;
    if ((rc2 = x.gather_rect(blocks[i].d_labels, bw, field_h, field_w, py, px, (uint32_t *)me.full_labels.p))) return rc2;
    if (me.rank != 0) return WS_OK;
    return gfail_if(g, me, ws_lists_from_arrival_device(me.ctx, merging, (const uint32_t *)me.full_keys.p, (const uint32_t *)me.full_labels.p, field_h, field_w,
                                                        n_seeds_total, opt, d_lakes, cap, n_lakes, offsets, uncoloured));
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

// Host buffers in and out.  Seeds: a list whose rows never decrease (every row-major sorted list; find_local_minima's) gives
// every rank ONE contiguous range, found by binary search; any other list is dealt out seed by seed with explicit colours
// (the general form).  With edge correction the rank's rows of the PADDED plane are built on the device: a zeroed block and
// a 2-D copy of the image rows that fall into it, one column in.
// the lake records of transform_to_list, wanted from a host-buffer tiled call (ws_transform_to_list_tiled)
struct HostLists {
  ws_lake *lakes;
  size_t cap;
  size_t *n_lakes;
  uint64_t *offsets, *uncoloured;
};

static int segment_tiled_host(ws_group *g, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc, size_t n_seeds,
                              const ws_options *opt, int merging, uint64_t *out_labels, uint32_t *exchange_rounds, const HostLists *lists) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if ((!img && h * w) || (!seeds_rc && n_seeds) || stride < w) return gfail(g, WS_ERR_BAD_ARG, "bad argument");
  const size_t e = opt->edge_correction ? 2 : 0, ph = h + e, pw = w + e, shift = opt->edge_correction && opt->seed_shift ? 1 : 0;
  if (!lists && !out_labels && ph * pw) return gfail(g, WS_ERR_BAD_ARG, "out_labels is null");
  if (n_seeds >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "too many seeds");
  if (ph > 0x7FFFFFF0ull || pw > 0x7FFFFFF0ull) return gfail(g, WS_ERR_TOO_LARGE, "plane too large");
  if (exchange_rounds) *exchange_rounds = 0;
  const int world = g->world;
  if (ph < (size_t)world) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row per rank");
  // the reference indexes the (padded) plane with the caller's coordinates and panics outside it (lib.rs:1675-1677)
  bool rows_sorted = true;
  for (size_t i = 0; i < n_seeds; ++i) {
    if (seeds_rc[2 * i] + shift >= ph || seeds_rc[2 * i + 1] + shift >= pw) return gfail(g, WS_ERR_SEED_OOB, "seed outside the label plane (the reference panics: lib.rs:1676)");
    if (i && seeds_rc[2 * i] < seeds_rc[2 * i - 2]) rows_sorted = false;
  }
  ws_options plain = *opt;
  plain.edge_correction = 0;      // the blocks hold the padded plane's own rows
  plain.seed_shift = 0;
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    G_HIP(g, hipSetDevice(me.device));
    hipStream_t s = me.ctx->stream;
    size_t r0, r1, lo, hi;
    int rc2 = ws_tile_rows(ph, me.rank, world, &r0, &r1, &lo, &hi);
    if (rc2) return gfail(g, rc2, "a field needs at least one row per rank");
    const size_t bh = hi - lo, bn = bh * pw;
    // ---- the rank's rows of the (padded) image
    if ((rc2 = grow(g, me.img, bn ? bn : 1))) return rc2;
    if ((rc2 = grow(g, me.labels, (bn ? bn : 1) * sizeof(uint32_t)))) return rc2;
    if (e) {
      G_HIP(g, hipMemsetAsync(me.img.p, 0, bn ? bn : 1, s));
      // padded row p holds image row p - 1: image rows [max(lo, 1) - 1, min(hi, h + 1) - 1)
      const size_t p0 = std::max<size_t>(lo, 1), p1 = std::min<size_t>(hi, h + 1);
      if (p1 > p0 && w)
        G_HIP(g, hipMemcpy2DAsync((uint8_t *)me.img.p + (p0 - lo) * pw + 1, pw, img + (p0 - 1) * stride, stride, w, p1 - p0, hipMemcpyHostToDevice, s));
    } else if (bn) {
      G_HIP(g, hipMemcpy2DAsync(me.img.p, pw, img + lo * stride, stride, w, bh, hipMemcpyHostToDevice, s));
    }
    // ---- its seeds, in local coordinates
    std::vector<uint32_t> loc, col;
    uint32_t first_colour = 1;
    bool explicit_colours = !rows_sorted;
    if (rows_sorted) {
      auto lower = [&](uint64_t row) {      // first list index whose (shifted) row is >= row
        size_t a = 0, b = n_seeds;
        while (a < b) { const size_t m = (a + b) / 2; if (seeds_rc[2 * m] + shift < row) a = m + 1; else b = m; }
        return a;
      };
      const size_t i0 = lower(lo), i1 = lower(hi);
      loc.resize(2 * (i1 - i0));
      for (size_t i = i0; i < i1; ++i) { loc[2 * (i - i0)] = (uint32_t)(seeds_rc[2 * i] + shift - lo); loc[2 * (i - i0) + 1] = (uint32_t)(seeds_rc[2 * i + 1] + shift); }
      first_colour = (uint32_t)i0 + 1u;
    } else {
      for (size_t i = 0; i < n_seeds; ++i) {
        const uint64_t row = seeds_rc[2 * i] + shift;
        if (row >= lo && row < hi) { loc.push_back((uint32_t)(row - lo)); loc.push_back((uint32_t)(seeds_rc[2 * i + 1] + shift)); col.push_back((uint32_t)i + 1u); }
      }
    }
    const size_t ns = loc.size() / 2;
    if ((rc2 = grow(g, me.seeds, (ns ? ns : 1) * 2 * sizeof(uint32_t)))) return rc2;
    if (ns) G_HIP(g, hipMemcpyAsync(me.seeds.p, loc.data(), ns * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (explicit_colours) {
      if ((rc2 = grow(g, me.colours, (ns ? ns : 1) * sizeof(uint32_t)))) return rc2;
      if (ns) G_HIP(g, hipMemcpyAsync(me.colours.p, col.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    }
    G_HIP(g, hipStreamSynchronize(s));      // loc / col leave scope; the block steps run on this stream anyway
    ws_tile_block b{};
    b.d_img = (const uint8_t *)me.img.p;
    b.d_seeds_rc = (const uint32_t *)me.seeds.p;
    b.d_colours = explicit_colours && world > 1 ? (const uint32_t *)me.colours.p : nullptr;
    b.n_seeds = ns;
    b.first_colour = first_colour;
    b.d_labels = (uint32_t *)me.labels.p;
    const size_t i = (size_t)(me.rank - g->first_local);
    if (world == 1) rc2 = single_rank(g, me, ph, pw, b, &plain, lists ? 0 : merging);
    else rc2 = tiled_rank(g, me, ph, pw, n_seeds, b, &plain, lists ? 0 : merging, &rounds[i]);
    if (rc2) return rc2;
    if (lists) {      // transform_to_list: the arrival planes of the owned rows to rank 0, the records of all levels there (see the device form)
      const uint32_t *keys = (const uint32_t *)me.keys.p, *full_k = keys, *full_l = b.d_labels;
      if (world == 1) {
        size_t kh = 0, kw = 0;
        G_WS(g, me, ws_last_arrival_device(me.ctx, &full_k, &kh, &kw));
      } else {
        Exchange x{g, me};
        const size_t n = ph * pw;
        if (me.rank == 0) {
          if ((rc2 = grow(g, me.full_keys, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
          if ((rc2 = grow(g, me.full_labels, (n ? n : 1) * sizeof(uint32_t)))) return rc2;
        }
        G_HIP(g, hipStreamSynchronize(s));
        if ((rc2 = x.gather_rows(keys + (r0 - lo) * pw, ph, pw, (uint32_t *)me.full_keys.p))) return rc2;
        if ((rc2 = x.gather_rows(b.d_labels + (r0 - lo) * pw, ph, pw, (uint32_t *)me.full_labels.p))) return rc2;
        full_k = (const uint32_t *)me.full_keys.p;
        full_l = (const uint32_t *)me.full_labels.p;
      }
      if (me.rank != 0) return WS_OK;
      if ((rc2 = grow(g, me.out64, std::max<size_t>(lists->cap, 1) * sizeof(ws_lake)))) return rc2;
      rc2 = ws_lists_from_arrival_device(me.ctx, merging, full_k, full_l, ph, pw, n_seeds, &plain, (ws_lake *)me.out64.p, lists->cap, lists->n_lakes,
                                         lists->offsets, lists->uncoloured);
      if (rc2 != WS_OK && rc2 != WS_ERR_CAPACITY) return gfail_if(g, me, rc2);
      const size_t got = std::min(*lists->n_lakes, lists->cap);
      // (many records: narrowed to u32 on the device -- a colour and an area of a plane below 2^31 pixels fit -- and widened into
      // the caller's ws_lake records by the host threads of rank 0's context, 8 M records a piece: half the bytes over the bus)
      const size_t piece_max = (size_t)1 << 23;
      if (got >= ((size_t)1 << 21) && ph * pw < 0x80000000ull) {
        int rc3;
        if ((rc3 = grow(g, me.rows, std::min(got, piece_max) * 2 * sizeof(uint32_t)))) return rc3;
        for (size_t off = 0; off < got; off += piece_max) {
          const size_t piece = std::min(piece_max, got - off);
          G_HIP(g, narrow_words(s, (const uint64_t *)((const ws_lake *)me.out64.p + off), (uint32_t *)me.rows.p, 2 * piece));
          G_WS(g, me, labels_to_host_u64(me.ctx, (const uint32_t *)me.rows.p, (uint64_t *)(lists->lakes + off), 2 * piece));
        }
      } else if (got) {
        G_HIP(g, hipMemcpyAsync(lists->lakes, me.out64.p, got * sizeof(ws_lake), hipMemcpyDeviceToHost, s));
      }
      G_HIP(g, hipStreamSynchronize(s));
      return gfail_if(g, me, rc2);
    }
    // ---- the rows it owns, widened, into the caller's plane
    const size_t own = (r1 - r0) * pw;
    // (as u32 chunks over the rank's own link, widened by host threads of the rank's context: ws_hostcopy.hip)
    if (own) G_WS(g, me, labels_to_host_u64(me.ctx, (const uint32_t *)me.labels.p + (r0 - lo) * pw, out_labels + r0 * pw, own));
    return WS_OK;
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

int ws_segment_tiled(ws_group *g, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc, size_t n_seeds,
                     const ws_options *opt, int merging, uint64_t *out_labels, uint32_t *exchange_rounds) {
  return segment_tiled_host(g, img, h, w, stride, seeds_rc, n_seeds, opt, merging, out_labels, exchange_rounds, nullptr);
}

// transform_to_list of a host field over the ranks of the group: as ws_segment_tiled up to the flood, then as
// ws_transform_to_list_tiled_device; the records go to the caller's host buffer from rank 0's device.
int ws_transform_to_list_tiled(ws_group *g, int merging, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc,
                               size_t n_seeds, const ws_options *opt, ws_lake *lakes, size_t cap, size_t *n_lakes, uint64_t *offsets,
                               uint64_t *uncoloured, uint32_t *exchange_rounds) {
  if (!g) return WS_ERR_BAD_ARG;
  if (!n_lakes || !offsets || !uncoloured || (!lakes && cap)) return gfail(g, WS_ERR_BAD_ARG, "null pointer");
  *n_lakes = 0;
  const HostLists lists{lakes, cap, n_lakes, offsets, uncoloured};
  return segment_tiled_host(g, img, h, w, stride, seeds_rc, n_seeds, opt, merging, nullptr, exchange_rounds, &lists);
}

// ... in py x px tiles: every local rank uploads its tile of the (padded) image, takes the seeds that fall on its plane with
// their colours (index + 1), runs tiled2d_rank and writes the rectangle it owns of out_labels.
int ws_segment_tiled2d(ws_group *g, const uint8_t *img, size_t h, size_t w, size_t stride, const uint64_t *seeds_rc, size_t n_seeds,
                       const ws_options *opt, int py, int px, int merging, uint64_t *out_labels, uint32_t *exchange_rounds) {
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if ((!img && h * w) || (!seeds_rc && n_seeds) || stride < w) return gfail(g, WS_ERR_BAD_ARG, "bad argument");
  if (py < 1 || px < 1 || py * px != g->world) return gfail(g, WS_ERR_BAD_ARG, "py * px must be the group's number of ranks");
  const size_t e = opt->edge_correction ? 2 : 0, ph = h + e, pw = w + e, shift = opt->edge_correction && opt->seed_shift ? 1 : 0;
  if (!out_labels && ph * pw) return gfail(g, WS_ERR_BAD_ARG, "out_labels is null");
  if (n_seeds >= 0x7FFFFFFFull) return gfail(g, WS_ERR_TOO_LARGE, "too many seeds");
  if (ph > 0x7FFFFFF0ull || pw > 0x7FFFFFF0ull) return gfail(g, WS_ERR_TOO_LARGE, "plane too large");
  if (ph < (size_t)py || pw < (size_t)px) return gfail(g, WS_ERR_BAD_ARG, "a field needs at least one row and one column per tile");
  if (exchange_rounds) *exchange_rounds = 0;
  for (size_t i = 0; i < n_seeds; ++i)      // the reference indexes the (padded) plane with the caller's coordinates and panics outside it (lib.rs:1675-1677)
    if (seeds_rc[2 * i] + shift >= ph || seeds_rc[2 * i + 1] + shift >= pw) return gfail(g, WS_ERR_SEED_OOB, "seed outside the label plane (the reference panics: lib.rs:1676)");
  ws_options plain = *opt;
  plain.edge_correction = 0;      // the tiles hold the padded plane's own pixels
  plain.seed_shift = 0;
  std::vector<uint32_t> rounds(g->ranks.size(), 0);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    G_HIP(g, hipSetDevice(me.device));
    hipStream_t s = me.ctx->stream;
    size_t rw[4], cl[4];
    int rc2 = ws_tile_grid(ph, pw, me.rank, py, px, rw, cl);
    if (rc2) return gfail(g, rc2, "bad tile grid");
    const size_t r0 = rw[0], r1 = rw[1], lo = rw[2], hi = rw[3], c0 = cl[0], c1 = cl[1], clo = cl[2], chi = cl[3];
    const size_t bh = hi - lo, bw = chi - clo, bn = bh * bw;
    if ((rc2 = grow(g, me.img, bn ? bn : 1))) return rc2;
    if ((rc2 = grow(g, me.labels, (bn ? bn : 1) * sizeof(uint32_t)))) return rc2;
    // padded pixel (p, q) holds image pixel (p - e / 2, q - e / 2); the ring of an edge-corrected plane is zero
    G_HIP(g, hipMemsetAsync(me.img.p, 0, bn ? bn : 1, s));
    const size_t o = e / 2;
    const size_t p0 = std::max(lo, o), p1 = std::min(hi, h + o), q0 = std::max(clo, o), q1 = std::min(chi, w + o);
    if (p1 > p0 && q1 > q0)
      G_HIP(g, hipMemcpy2DAsync((uint8_t *)me.img.p + (p0 - lo) * bw + (q0 - clo), bw, img + (p0 - o) * stride + (q0 - o), stride, q1 - q0, p1 - p0, hipMemcpyHostToDevice, s));
    std::vector<uint32_t> loc, col;
    for (size_t i = 0; i < n_seeds; ++i) {
      const uint64_t row = seeds_rc[2 * i] + shift, cc = seeds_rc[2 * i + 1] + shift;
      if (row >= lo && row < hi && cc >= clo && cc < chi) { loc.push_back((uint32_t)(row - lo)); loc.push_back((uint32_t)(cc - clo)); col.push_back((uint32_t)i + 1u); }
    }
    const size_t ns = col.size();
    if ((rc2 = grow(g, me.seeds, (ns ? ns : 1) * 2 * sizeof(uint32_t)))) return rc2;
    if ((rc2 = grow(g, me.colours, (ns ? ns : 1) * sizeof(uint32_t)))) return rc2;
    if (ns) G_HIP(g, hipMemcpyAsync(me.seeds.p, loc.data(), ns * 2 * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    if (ns) G_HIP(g, hipMemcpyAsync(me.colours.p, col.data(), ns * sizeof(uint32_t), hipMemcpyHostToDevice, s));
    G_HIP(g, hipStreamSynchronize(s));
    ws_tile_block2d b{};
    b.d_img = (const uint8_t *)me.img.p;
    b.img_stride = bw;
    b.d_seeds_rc = (const uint32_t *)me.seeds.p;
    b.d_colours = (const uint32_t *)me.colours.p;
    b.n_seeds = ns;
    b.d_labels = (uint32_t *)me.labels.p;
    const size_t i = (size_t)(me.rank - g->first_local);
    if ((rc2 = tiled2d_rank(g, me, ph, pw, py, px, n_seeds, b, &plain, merging, &rounds[i]))) return rc2;
    // ---- the rectangle it owns, widened, into the caller's plane
    const size_t oh = r1 - r0, ow = c1 - c0;
    if (oh * ow && host_copy_in_chunks(me.ctx, oh * ow)) {
      // the rectangle packed on the device, over the bus as u32 chunks, widened into its rows of the caller's plane (ws_hostcopy.hip)
      if ((rc2 = grow(g, me.rows, oh * ow * sizeof(uint32_t)))) return rc2;
      G_HIP(g, hipMemcpy2DAsync(me.rows.p, ow * sizeof(uint32_t), (const uint32_t *)me.labels.p + (r0 - lo) * bw + (c0 - clo), bw * sizeof(uint32_t),
                                ow * sizeof(uint32_t), oh, hipMemcpyDeviceToDevice, s));
      G_WS(g, me, labels_to_host_u64(me.ctx, (const uint32_t *)me.rows.p, out_labels + r0 * pw + c0, oh * ow, nullptr, ow, pw));
    } else if (oh * ow) {
      if ((rc2 = grow(g, me.out64, bn * sizeof(uint64_t)))) return rc2;
      G_HIP(g, widen_labels(s, (const uint32_t *)me.labels.p, (uint64_t *)me.out64.p, bn));
      G_HIP(g, hipMemcpy2DAsync(out_labels + r0 * pw + c0, pw * sizeof(uint64_t), (const uint64_t *)me.out64.p + (r0 - lo) * bw + (c0 - clo), bw * sizeof(uint64_t),
                                ow * sizeof(uint64_t), oh, hipMemcpyDeviceToHost, s));
      G_HIP(g, hipStreamSynchronize(s));
    }
    return WS_OK;
  });
  if (exchange_rounds) *exchange_rounds = rounds[0];
  return rc;
}

int ws_segment_batch_group(ws_group *g, size_t h, size_t w, const ws_batch_part *parts, const ws_options *opt, size_t *failed_rank,
                           size_t *failed_slice) {
  if (failed_rank) *failed_rank = 0;
  if (failed_slice) *failed_slice = 0;
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (!parts) return gfail(g, WS_ERR_BAD_ARG, "parts pointer is null");
  std::vector<size_t> bad(g->ranks.size(), 0);
  std::vector<int> rcs(g->ranks.size(), WS_OK);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    const size_t i = (size_t)(me.rank - g->first_local);
    const ws_batch_part &p = parts[i];
    if (p.n_slices == 0) return WS_OK;
    rcs[i] = ws_segment_batch_device(me.ctx, p.d_cube, p.n_slices, h, w, w, h * w, p.d_seeds_rc, p.seed_offsets, opt, p.d_labels, &bad[i]);
    if (rcs[i] != WS_OK) return gfail(g, rcs[i], std::string("rank ") + std::to_string(me.rank) + ": ws_segment_batch_device: " + ws_last_error(me.ctx));
    return WS_OK;
  });
  for (size_t i = 0; i < rcs.size(); ++i)
    if (rcs[i] != WS_OK) { if (failed_rank) *failed_rank = (size_t)g->first_local + i; if (failed_slice) *failed_slice = bad[i]; break; }
  return rc;
}

// A cube of independent slices in HOST memory over the ranks of a group: rank r takes the slices [n r / world, n (r + 1) / world)
// and runs them as ws_segment_batch on its own context -- four lanes per device, every device on its own link to the host.
// A process drives its local ranks only (an RCCL group: its one rank's block of slices; the cube pointer is that process's).
int ws_segment_batch_host(ws_group *g, const uint8_t *cube, size_t n_slices, size_t h, size_t w, size_t row_stride, size_t slice_stride,
                          const uint64_t *seeds_rc, const size_t *seed_offsets, const ws_options *opt, uint64_t *out_labels,
                          size_t *n_seeds, size_t *failed_slice) {
  if (failed_slice) *failed_slice = 0;
  int rc = check_group_call(g, opt);
  if (rc) return rc;
  if (n_slices == 0) return WS_OK;
  if (!out_labels || (!cube && h * w) || (seeds_rc && !seed_offsets)) return gfail(g, WS_ERR_BAD_ARG, "null pointer");
  const size_t e = opt->edge_correction ? 2 : 0, plane = (h + e) * (w + e);
  const size_t world = (size_t)g->world;
  std::vector<size_t> bad(g->ranks.size(), 0);
  std::vector<int> rcs(g->ranks.size(), WS_OK);
  rc = for_local_ranks(g, [&](Rank &me) -> int {
    const size_t i = (size_t)(me.rank - g->first_local);
    const size_t k0 = n_slices * (size_t)me.rank / world, k1 = n_slices * ((size_t)me.rank + 1) / world;
    if (k1 == k0) return WS_OK;
    // (the rank's seed offsets as they are: ws_segment_batch only looks at differences and at seeds_rc + 2 * offset)
    rcs[i] = ws_segment_batch(me.ctx, cube + k0 * slice_stride, k1 - k0, h, w, row_stride, slice_stride, seeds_rc, seeds_rc ? seed_offsets + k0 : nullptr,
                              opt, out_labels + k0 * plane, n_seeds ? n_seeds + k0 : nullptr, &bad[i]);
    bad[i] += k0;
    if (rcs[i] != WS_OK) return gfail(g, rcs[i], std::string("rank ") + std::to_string(me.rank) + ": ws_segment_batch: " + ws_last_error(me.ctx));
    return WS_OK;
  });
  for (size_t i = 0; i < rcs.size(); ++i)      // ranks hold increasing blocks of slices: the first failing rank has the lowest failing slice
    if (rcs[i] != WS_OK) { if (failed_slice) *failed_slice = bad[i]; break; }
  return rc;
}

}  // extern "C"
