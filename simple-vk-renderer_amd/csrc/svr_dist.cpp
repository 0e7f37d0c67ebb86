// svr_dist.cpp — include/svr_dist.h: the sharded frame for a C++ host, over raw RCCL (or, for tests on a
// one-GPU box, a shared-memory transport).  Host code only; it sits ABOVE the C ABI of include/svr.h and
// uses nothing of the renderer but its exported entry points.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/svr_dist.h"

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                       \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return fail(SVR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define NCCLCHK(expr)                                                                                        \
  do {                                                                                                       \
    ncclResult_t e_ = (expr);                                                                                \
    if (e_ != ncclSuccess) return fail(SVR_ERR_DEVICE, std::string(#expr) + ": " + ncclGetErrorString(e_)); \
  } while (0)
#define SVRCHK(expr)                                                              \
  do {                                                                            \
    int e_ = (expr);                                                              \
    if (e_ != SVR_OK) return fail(e_, std::string(#expr) + ": " + svr_last_error()); \
  } while (0)

// shared-memory transport (tests): header + one image per frame slot + one cost profile per rank
struct ShmHeader {
  std::atomic<uint32_t> ready;
  std::atomic<uint32_t> abort;    // a rank failed: nobody waits for it any longer
  std::atomic<uint32_t> arrived;  // sense-reversing barrier: ranks that reached the current round ...
  std::atomic<uint32_t> round;    // ... and the round's number
  uint32_t status[2][64];         // per frame slot and rank: the present's status word (svr_set_present_status)
};

// barrier over the shared region.  A pthread barrier would leave the peers of a rank that died or bailed out blocked
// for ever (a test then hangs until its timeout instead of failing with the error): this one gives up when a rank
// raised the abort flag, or after a minute.
int shm_barrier(ShmHeader* h, int world) {
  const uint32_t my_round = h->round.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->round.store(my_round + 1, std::memory_order_release);
    return SVR_OK;
  }
  for (uint64_t spin = 0; h->round.load(std::memory_order_acquire) == my_round; spin++) {
    if (h->abort.load(std::memory_order_acquire)) return fail(SVR_ERR_DEVICE, "shared-memory transport: a peer failed");
    if (spin > 1200000) {  // 60 s
      h->abort.store(1, std::memory_order_release);
      return fail(SVR_ERR_DEVICE, "shared-memory transport: a peer did not arrive within 60 s");
    }
    usleep(50);
  }
  return SVR_OK;
}

// Bottleneck-optimal cut of `rows` non-negative costs into `world` consecutive bands: binary search on the
// largest band's cost over the prefix sums, greedy feasibility (every band takes as many rows as fit).
// Integer arithmetic only: every rank computes the same boundaries from the same profile (dist.py balanced_bounds).
std::vector<uint32_t> balanced_bounds(const std::vector<uint64_t>& cost, int world) {
  const size_t h = cost.size();
  std::vector<uint64_t> pre(h + 1, 0);
  uint64_t biggest = 0;
  for (size_t i = 0; i < h; i++) {
    pre[i + 1] = pre[i] + cost[i];
    biggest = std::max(biggest, cost[i]);
  }
  auto cut = [&](uint64_t limit, std::vector<uint32_t>* out) {
    size_t b = 0;
    if (out) out->assign(1, 0u);
    for (int r = 0; r < world; r++) {
      // largest e with pre[e] <= pre[b] + limit
      size_t e = (size_t)(std::upper_bound(pre.begin(), pre.end(), pre[b] + limit) - pre.begin()) - 1;
      e = std::min(std::max(e, b), h);
      b = e;
      if (out) out->push_back((uint32_t)b);
    }
    return b >= h;
  };
  uint64_t lo = biggest, hi = pre[h];
  while (lo < hi) {
    uint64_t mid = lo + (hi - lo) / 2;
    if (cut(mid, nullptr)) hi = mid;
    else lo = mid + 1;
  }
  std::vector<uint32_t> b;
  cut(lo, &b);
  b[0] = 0;
  b[(size_t)world] = (uint32_t)h;
  return b;
}

std::vector<uint32_t> equal_bounds(uint32_t height, int world) {
  const uint32_t band = (height + (uint32_t)world - 1) / (uint32_t)world;
  std::vector<uint32_t> b;
  for (int r = 0; r < world; r++) b.push_back(std::min((uint32_t)r * band, height));
  b.push_back(height);
  return b;
}

}  // namespace

struct SvrDist {
  SvrContext* ctx = nullptr;
  int transport = SVR_DIST_RCCL, rank = 0, world = 1, fmt = SVR_SWAPCHAIN_B8G8R8A8;
  uint32_t W = 0, H = 0, band = 0;  // band: rows of an equal band (the images are padded to band * world rows, at least)
  uint32_t rows_padded = 0;         // ... and to whole groups of `world` tile rows (the interleaved partition's all-gathers)
  int partition = SVR_DIST_BANDS;   // what the next begin_frame uses
  uint32_t replays = 0;             // frames exchanged a second time because a rank's present had been void
  std::vector<uint32_t> bounds;
  hipStream_t render = nullptr, comm = nullptr;
  static const int SLOTS = 2;
  struct Slot {
    void* color = nullptr;      // RGBA16F _draw_image, padded height
    float* depth = nullptr;
    uint8_t* image = nullptr;   // the swapchain image that travels
    hipEvent_t rendered = nullptr, exchanged = nullptr;
    std::vector<uint32_t> bounds;  // the partition this slot's frame in flight was rendered with
    int partition = SVR_DIST_BANDS;
    uint32_t* status = nullptr;    // device, [world]: every rank's present status of this frame (own word: [rank])
    uint32_t* h_status = nullptr;  // pinned, [world]: the same after the exchange
    bool in_flight = false;
  } slots[SLOTS];
  int cur = -1;                 // slot between begin_frame and end_frame
  std::vector<int> fifo;        // slots whose exchange has been started, oldest first
  // RCCL
  ncclComm_t comm_rccl = nullptr;
  uint64_t* d_profile = nullptr;  // [H] device staging of the all-reduced cost profile
  // SHM
  std::string shm_name;
  void* shm = nullptr;
  size_t shm_bytes = 0;
  ShmHeader* hdr = nullptr;
  uint8_t* shm_images = nullptr;
  uint64_t* shm_profiles = nullptr;
};

namespace {

size_t padded_rows(const SvrDist* d) { return d->rows_padded; }
size_t image_bytes(const SvrDist* d) { return padded_rows(d) * d->W * 4; }

// rows [first, first + n) a rank contributes to a frame: one run per band, one per tile row when they are interleaved
struct RowRun {
  uint32_t first, n;
};
std::vector<RowRun> runs_of(const SvrDist* d, const SvrDist::Slot& s, int rank) {
  std::vector<RowRun> out;
  if (s.partition == SVR_DIST_INTERLEAVED) {
    for (uint32_t t = (uint32_t)rank; t * 32u < d->H; t += (uint32_t)d->world) out.push_back({t * 32u, std::min(32u, d->H - t * 32u)});
  } else {
    const uint32_t y0 = s.bounds[(size_t)rank], n = s.bounds[(size_t)rank + 1] - y0;
    if (n) out.push_back({y0, n});
  }
  return out;
}

int exchange(SvrDist* d, SvrDist::Slot& s) {
  const size_t row_bytes = (size_t)d->W * 4;
  const std::vector<uint32_t>& b = s.bounds;
  if (d->transport == SVR_DIST_RCCL) {  // (also with one rank: the same calls, trivially)
    HIPCHK(hipStreamWaitEvent(d->comm, s.rendered, 0));
    // A failing call inside a group must not leave the group open (every later RCCL call of this thread would queue
    // into it): the first error is kept, the group is always ended.
    ncclResult_t first_err = ncclSuccess;
    const char* what = "";
    auto note = [&](ncclResult_t r, const char* w) {
      if (r != ncclSuccess && first_err == ncclSuccess) {
        first_err = r;
        what = w;
      }
    };
    if (s.partition == SVR_DIST_INTERLEAVED) {
      // tile row t belongs to rank t % world: every group of `world` consecutive tile rows is one in-place all-gather
      // of 32-row chunks (the image is padded to whole groups), all of them in one RCCL group = one launch
      const size_t chunk = 32 * row_bytes;
      note(ncclGroupStart(), "ncclGroupStart");
      for (size_t g = 0; g * (size_t)d->world * 32 < d->H; g++) {
        uint8_t* base = s.image + g * (size_t)d->world * chunk;
        note(ncclAllGather(base + (size_t)d->rank * chunk, base, chunk, ncclUint8, d->comm_rccl, d->comm), "ncclAllGather");
      }
      note(ncclGroupEnd(), "ncclGroupEnd");
    } else if (b == equal_bounds(d->H, d->world)) {  // one in-place all-gather of equal chunks (the image is padded to band * world rows)
      const size_t n = (size_t)d->band * row_bytes;
      note(ncclAllGather(s.image + (size_t)d->rank * n, s.image, n, ncclUint8, d->comm_rccl, d->comm), "ncclAllGather");
    } else {  // unequal bands: every band straight into its rows of every peer's image, one group
      note(ncclGroupStart(), "ncclGroupStart");
      const size_t mine = (size_t)(b[(size_t)d->rank + 1] - b[(size_t)d->rank]) * row_bytes;
      for (int peer = 0; peer < d->world; peer++) {
        if (peer == d->rank) continue;
        if (mine) note(ncclSend(s.image + (size_t)b[(size_t)d->rank] * row_bytes, mine, ncclUint8, peer, d->comm_rccl, d->comm), "ncclSend");
        const size_t theirs = (size_t)(b[(size_t)peer + 1] - b[(size_t)peer]) * row_bytes;
        if (theirs) note(ncclRecv(s.image + (size_t)b[(size_t)peer] * row_bytes, theirs, ncclUint8, peer, d->comm_rccl, d->comm), "ncclRecv");
      }
      note(ncclGroupEnd(), "ncclGroupEnd");
    }
    // every rank's present status travels behind the rows: a rank whose pass overflowed sent stale rows, and everybody
    // learns it with the frame (svr_dist_wait_frame)
    note(ncclAllGather(s.status + d->rank, s.status, 1, ncclUint32, d->comm_rccl, d->comm), "ncclAllGather(status)");
    if (first_err != ncclSuccess) return fail(SVR_ERR_DEVICE, std::string(what) + ": " + ncclGetErrorString(first_err));
    HIPCHK(hipMemcpyAsync(s.h_status, s.status, (size_t)d->world * 4, hipMemcpyDeviceToHost, d->comm));
    HIPCHK(hipEventRecord(s.exchanged, d->comm));
    return SVR_OK;
  }
  if (d->world == 1) {
    HIPCHK(hipMemcpyAsync(s.h_status, s.status, 4, hipMemcpyDeviceToHost, d->render));
    HIPCHK(hipEventRecord(s.exchanged, d->render));
    return SVR_OK;
  }
  // shared memory (tests): blocking.  One barrier behind the reads is enough: a rank enters the next exchange only
  // after it has read everything it wanted from this one.
  const int slot = (int)(&s - d->slots);
  uint8_t* shared = d->shm_images + (size_t)slot * image_bytes(d);
  auto bail = [&](int code) {  // the peers must not wait for a rank that gave up
    d->hdr->abort.store(1, std::memory_order_release);
    return code;
  };
  if (hipEventSynchronize(s.rendered) != hipSuccess) return bail(fail(SVR_ERR_DEVICE, "hipEventSynchronize(rendered)"));
  for (const RowRun& r : runs_of(d, s, d->rank))
    if (hipMemcpy(shared + r.first * row_bytes, s.image + r.first * row_bytes, r.n * row_bytes, hipMemcpyDeviceToHost) != hipSuccess)
      return bail(fail(SVR_ERR_DEVICE, "hipMemcpy of the band to the shared image"));
  if (hipMemcpy(&d->hdr->status[slot][d->rank], s.status + d->rank, 4, hipMemcpyDeviceToHost) != hipSuccess)
    return bail(fail(SVR_ERR_DEVICE, "hipMemcpy of the present status"));
  if (int e = shm_barrier(d->hdr, d->world)) return e;
  for (int peer = 0; peer < d->world; peer++) {
    s.h_status[peer] = d->hdr->status[slot][peer];
    if (peer == d->rank) continue;
    for (const RowRun& r : runs_of(d, s, peer))
      if (hipMemcpy(s.image + r.first * row_bytes, shared + r.first * row_bytes, r.n * row_bytes, hipMemcpyHostToDevice) != hipSuccess)
        return bail(fail(SVR_ERR_DEVICE, "hipMemcpy of a peer's band"));
  }
  if (int e = shm_barrier(d->hdr, d->world)) return e;  // nobody overwrites the shared image while a peer still reads it
  HIPCHK(hipEventRecord(s.exchanged, d->comm));
  return SVR_OK;
}

}  // namespace

extern "C" {

const char* svr_dist_last_error(void) { return g_err.c_str(); }

int svr_dist_get_unique_id(int transport, uint8_t id[SVR_DIST_ID_BYTES]) {
  if (!id) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_get_unique_id: null argument");
  std::memset(id, 0, SVR_DIST_ID_BYTES);
  if (transport == SVR_DIST_RCCL) {
    static_assert(sizeof(ncclUniqueId) <= SVR_DIST_ID_BYTES, "ncclUniqueId");
    ncclUniqueId nid;
    NCCLCHK(ncclGetUniqueId(&nid));
    std::memcpy(id, &nid, sizeof(nid));
    return SVR_OK;
  }
  if (transport == SVR_DIST_SHM) {
    snprintf(reinterpret_cast<char*>(id), SVR_DIST_ID_BYTES, "/svr_dist_%d_%u", (int)getpid(), (unsigned)rand());
    return SVR_OK;
  }
  return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_get_unique_id: unknown transport");
}

int svr_dist_create(SvrContext* ctx, int transport, const uint8_t id[SVR_DIST_ID_BYTES], int rank, int world, uint32_t width,
                    uint32_t height, int swapchain_format, SvrDist** out) {
  if (!ctx || !id || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_create: null argument");
  if (world < 1 || rank < 0 || rank >= world || world > 64) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_create: bad rank / world");
  if (width == 0 || height == 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_create: empty extent");
  if (transport != SVR_DIST_RCCL && transport != SVR_DIST_SHM) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_create: unknown transport");
  SvrDist* d = new SvrDist();
  d->ctx = ctx;
  d->transport = transport;
  d->rank = rank;
  d->world = world;
  d->W = width;
  d->H = height;
  d->fmt = swapchain_format;
  d->band = (height + (uint32_t)world - 1) / (uint32_t)world;
  {
    const uint32_t tile_rows = (height + 31u) / 32u, groups = (tile_rows + (uint32_t)world - 1) / (uint32_t)world;
    d->rows_padded = std::max(d->band * (uint32_t)world, groups * (uint32_t)world * 32u);
  }
  d->bounds = equal_bounds(height, world);
  auto bail = [&](int code) {
    svr_dist_destroy(d);
    return code;
  };
#define TRY_HIP(expr)                                                                                      \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) return bail(fail(SVR_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_))); \
  } while (0)
  TRY_HIP(hipStreamCreateWithFlags(&d->render, hipStreamNonBlocking));
  TRY_HIP(hipStreamCreateWithFlags(&d->comm, hipStreamNonBlocking));
  const size_t rows = padded_rows(d);
  for (auto& s : d->slots) {
    TRY_HIP(hipMalloc(&s.color, rows * width * 8));
    TRY_HIP(hipMalloc((void**)&s.depth, rows * width * 4));
    TRY_HIP(hipMalloc((void**)&s.image, rows * width * 4));
    TRY_HIP(hipMemset(s.color, 0, rows * width * 8));
    TRY_HIP(hipMemset(s.depth, 0, rows * width * 4));
    TRY_HIP(hipMemset(s.image, 0, rows * width * 4));
    TRY_HIP(hipMalloc((void**)&s.status, 64 * sizeof(uint32_t)));
    TRY_HIP(hipMemset(s.status, 0, 64 * sizeof(uint32_t)));
    TRY_HIP(hipHostMalloc((void**)&s.h_status, 64 * sizeof(uint32_t), hipHostMallocDefault));
    std::memset(s.h_status, 0, 64 * sizeof(uint32_t));
    TRY_HIP(hipEventCreateWithFlags(&s.rendered, hipEventDisableTiming));
    TRY_HIP(hipEventCreateWithFlags(&s.exchanged, hipEventDisableTiming));
  }
  TRY_HIP(hipMalloc((void**)&d->d_profile, (size_t)height * 8));
  if (svr_set_stream(ctx, d->render) != SVR_OK) return bail(fail(SVR_ERR_DEVICE, std::string("svr_set_stream: ") + svr_last_error()));
  if (transport == SVR_DIST_RCCL) {
    ncclUniqueId nid;
    std::memcpy(&nid, id, sizeof(nid));
    ncclResult_t r = ncclCommInitRank(&d->comm_rccl, world, nid, rank);
    if (r != ncclSuccess) return bail(fail(SVR_ERR_DEVICE, std::string("ncclCommInitRank: ") + ncclGetErrorString(r)));
  } else {
    d->shm_name = reinterpret_cast<const char*>(id);
    d->shm_bytes = 4096 + (size_t)SvrDist::SLOTS * image_bytes(d) + (size_t)world * height * 8;
    int fd = shm_open(d->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    const bool creator = fd >= 0;
    if (!creator) fd = shm_open(d->shm_name.c_str(), O_RDWR, 0600);
    if (fd < 0) return bail(fail(SVR_ERR_DEVICE, "svr_dist_create: shm_open failed"));
    if (creator && ftruncate(fd, (off_t)d->shm_bytes) != 0) {
      close(fd);
      return bail(fail(SVR_ERR_DEVICE, "svr_dist_create: ftruncate failed"));
    }
    if (!creator) {  // wait until the creator has sized the object
      struct stat st;
      for (int spin = 0; spin < 200000; spin++) {
        if (fstat(fd, &st) == 0 && (size_t)st.st_size >= d->shm_bytes) break;
        usleep(50);
      }
    }
    d->shm = mmap(nullptr, d->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (d->shm == MAP_FAILED) {
      d->shm = nullptr;
      return bail(fail(SVR_ERR_DEVICE, "svr_dist_create: mmap failed"));
    }
    d->hdr = reinterpret_cast<ShmHeader*>(d->shm);
    d->shm_images = reinterpret_cast<uint8_t*>(d->shm) + 4096;
    d->shm_profiles = reinterpret_cast<uint64_t*>(d->shm_images + (size_t)SvrDist::SLOTS * image_bytes(d));
    if (creator) {  // (ftruncate zeroed the region: abort, arrived, round and the status words start at 0)
      d->hdr->ready.store(1u, std::memory_order_release);
    } else {
      for (int spin = 0; spin < 200000 && d->hdr->ready.load(std::memory_order_acquire) == 0u; spin++) usleep(50);
      if (d->hdr->ready.load(std::memory_order_acquire) == 0u) return bail(fail(SVR_ERR_DEVICE, "svr_dist_create: shared region never became ready"));
    }
    const int arrived = shm_barrier(d->hdr, world);  // everyone has mapped it: the name can go
    if (creator) shm_unlink(d->shm_name.c_str());
    if (arrived != SVR_OK) return bail(arrived);
  }
#undef TRY_HIP
  *out = d;
  return SVR_OK;
}

void svr_dist_destroy(SvrDist* d) {
  if (!d) return;
  if (d->ctx) {
    (void)svr_sync(d->ctx);
    (void)svr_bind_targets(d->ctx, nullptr, nullptr);
    (void)svr_set_scissor(d->ctx, 0, 0, d->W, d->H);
    (void)svr_set_row_interleave(d->ctx, 1, 0);
    (void)svr_set_present_status(d->ctx, nullptr);
    (void)svr_set_stream(d->ctx, nullptr);
  }
  if (d->comm) (void)hipStreamSynchronize(d->comm);
  if (d->comm_rccl) (void)ncclCommDestroy(d->comm_rccl);
  for (auto& s : d->slots) {
    if (s.color) (void)hipFree(s.color);
    if (s.depth) (void)hipFree(s.depth);
    if (s.image) (void)hipFree(s.image);
    if (s.status) (void)hipFree(s.status);
    if (s.h_status) (void)hipHostFree(s.h_status);
    if (s.rendered) (void)hipEventDestroy(s.rendered);
    if (s.exchanged) (void)hipEventDestroy(s.exchanged);
  }
  if (d->d_profile) (void)hipFree(d->d_profile);
  if (d->render) (void)hipStreamDestroy(d->render);
  if (d->comm) (void)hipStreamDestroy(d->comm);
  if (d->shm) munmap(d->shm, d->shm_bytes);
  delete d;
}

int svr_dist_get_bounds(SvrDist* d, uint32_t* bounds, size_t capacity) {
  if (!d || !bounds) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_get_bounds: null argument");
  if (capacity < d->bounds.size()) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_get_bounds: buffer too small");
  std::copy(d->bounds.begin(), d->bounds.end(), bounds);
  return SVR_OK;
}

int svr_dist_set_bounds(SvrDist* d, const uint32_t* bounds, size_t count) {
  if (!d || !bounds) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_set_bounds: null argument");
  if (count != (size_t)d->world + 1 || bounds[0] != 0 || bounds[count - 1] != d->H)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_set_bounds: need world + 1 values from 0 to height");
  for (size_t i = 0; i + 1 < count; i++)
    if (bounds[i] > bounds[i + 1]) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_set_bounds: boundaries must not decrease");
  d->bounds.assign(bounds, bounds + count);
  return SVR_OK;
}

int svr_dist_band(SvrDist* d, uint32_t* first_row, uint32_t* rows) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_band: null argument");
  if (first_row) *first_row = d->bounds[(size_t)d->rank];
  if (rows) *rows = d->bounds[(size_t)d->rank + 1] - d->bounds[(size_t)d->rank];
  return SVR_OK;
}

int svr_dist_rebalance(SvrDist* d, float measured_gpu_ms, int* changed) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_rebalance: null argument");
  if (changed) *changed = 0;
  if (d->world == 1 || d->partition == SVR_DIST_INTERLEAVED) return SVR_OK;  // interleaved rows are balanced by construction
  // this rank's tile-row costs, spread over the pixel rows they cover (x1024 keeps the remainders), then scaled so
  // that the band adds up to its measured time in nanoseconds
  std::vector<uint32_t> costs(512);
  uint32_t n = 0, y0 = 0, rows = 0;
  SVRCHK(svr_get_row_costs(d->ctx, costs.data(), costs.size(), &n, &y0, &rows));
  std::vector<uint64_t> mine(d->H, 0);
  uint64_t total = 0;
  for (uint32_t t = 0; t < n; t++) {
    const uint32_t a = y0 + 32u * t, b = std::min(a + 32u, y0 + rows);
    for (uint32_t y = a; y < b && y < d->H; y++) {
      mine[y] = ((uint64_t)costs[t] * 1024u) / (b - a);
      total += mine[y];
    }
  }
  if (measured_gpu_ms > 0.f && total) {
    const uint64_t ns = (uint64_t)((double)measured_gpu_ms * 1e6);
    for (auto& v : mine) v = v * ns / total;
  }
  std::vector<uint64_t> profile(d->H, 0);
  if (d->transport == SVR_DIST_RCCL) {
    HIPCHK(hipMemcpyAsync(d->d_profile, mine.data(), (size_t)d->H * 8, hipMemcpyHostToDevice, d->comm));
    NCCLCHK(ncclAllReduce(d->d_profile, d->d_profile, d->H, ncclUint64, ncclSum, d->comm_rccl, d->comm));
    HIPCHK(hipMemcpyAsync(profile.data(), d->d_profile, (size_t)d->H * 8, hipMemcpyDeviceToHost, d->comm));
    HIPCHK(hipStreamSynchronize(d->comm));
  } else {
    std::memcpy(d->shm_profiles + (size_t)d->rank * d->H, mine.data(), (size_t)d->H * 8);
    if (int e = shm_barrier(d->hdr, d->world)) return e;
    for (int r = 0; r < d->world; r++)
      for (uint32_t y = 0; y < d->H; y++) profile[y] += d->shm_profiles[(size_t)r * d->H + y];
    if (int e = shm_barrier(d->hdr, d->world)) return e;
  }
  uint64_t sum = 0, covered = 0;
  for (uint64_t v : profile) {
    sum += v;
    covered += v ? 1 : 0;
  }
  if (sum == 0) return SVR_OK;
  if (covered < profile.size()) {  // rows nobody reported yet count as average rows
    const uint64_t avg = std::max<uint64_t>(1, sum / covered);
    for (auto& v : profile)
      if (!v) v = avg;
  }
  std::vector<uint32_t> cut = balanced_bounds(profile, d->world);
  if (changed) *changed = cut != d->bounds;
  d->bounds.swap(cut);
  return SVR_OK;
}

int svr_dist_set_partition(SvrDist* d, int partition) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_set_partition: null argument");
  if (partition != SVR_DIST_BANDS && partition != SVR_DIST_INTERLEAVED) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_set_partition: unknown partition");
  d->partition = partition;
  return SVR_OK;
}

int svr_dist_get_partition(SvrDist* d, int* partition) {
  if (!d || !partition) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_get_partition: null argument");
  *partition = d->partition;
  return SVR_OK;
}

int svr_dist_pick_partition(SvrDist* d, float bands_ms, float interleaved_ms, int* picked) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_pick_partition: null argument");
  // the frame is as slow as its slowest rank: max over the ranks of either time, then the smaller of the two
  uint64_t mine[2] = {(uint64_t)(std::max(bands_ms, 0.f) * 1e6), (uint64_t)(std::max(interleaved_ms, 0.f) * 1e6)}, all[2] = {mine[0], mine[1]};
  if (d->world > 1) {
    if (d->transport == SVR_DIST_RCCL) {
      HIPCHK(hipMemcpyAsync(d->d_profile, mine, 16, hipMemcpyHostToDevice, d->comm));
      NCCLCHK(ncclAllReduce(d->d_profile, d->d_profile, 2, ncclUint64, ncclMax, d->comm_rccl, d->comm));
      HIPCHK(hipMemcpyAsync(all, d->d_profile, 16, hipMemcpyDeviceToHost, d->comm));
      HIPCHK(hipStreamSynchronize(d->comm));
    } else {
      std::memcpy(d->shm_profiles + (size_t)d->rank * d->H, mine, 16);
      if (int e = shm_barrier(d->hdr, d->world)) return e;
      for (int r = 0; r < d->world; r++)
        for (int k = 0; k < 2; k++) all[k] = std::max(all[k], d->shm_profiles[(size_t)r * d->H + (size_t)k]);
      if (int e = shm_barrier(d->hdr, d->world)) return e;
    }
  }
  d->partition = all[1] < all[0] ? SVR_DIST_INTERLEAVED : SVR_DIST_BANDS;
  if (picked) *picked = d->partition;
  return SVR_OK;
}

int svr_dist_replays(SvrDist* d, uint32_t* n) {
  if (!d || !n) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_replays: null argument");
  *n = d->replays;
  return SVR_OK;
}

int svr_dist_begin_frame(SvrDist* d) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_begin_frame: null argument");
  if (d->cur >= 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_begin_frame: the previous frame was not ended");
  int slot = -1;
  for (int i = 0; i < SvrDist::SLOTS; i++)
    if (!d->slots[i].in_flight) {
      slot = i;
      break;
    }
  if (slot < 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_begin_frame: both frame slots are in flight (svr_dist_wait_frame first)");
  SvrDist::Slot& s = d->slots[slot];
  s.bounds = d->bounds;
  s.partition = d->partition;
  SVRCHK(svr_bind_targets(d->ctx, s.color, s.depth));
  if (s.partition == SVR_DIST_INTERLEAVED) {
    SVRCHK(svr_set_scissor(d->ctx, 0, 0, d->W, d->H));
    SVRCHK(svr_set_row_interleave(d->ctx, (uint32_t)d->world, (uint32_t)d->rank));
  } else {
    SVRCHK(svr_set_row_interleave(d->ctx, 1, 0));
    const uint32_t y0 = s.bounds[(size_t)d->rank], rows = s.bounds[(size_t)d->rank + 1] - y0;
    if (rows) SVRCHK(svr_set_scissor(d->ctx, 0, y0, d->W, rows));
  }
  d->cur = slot;
  return SVR_OK;
}

int svr_dist_end_frame(SvrDist* d) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_end_frame: null argument");
  if (d->cur < 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_end_frame: no frame was begun");
  SvrDist::Slot& s = d->slots[d->cur];
  // vkutil::copy_image of this rank's rows: identity extent = the scissor's rows / the context's own tile rows
  // (include/svr.h); the present reports into the slot's status word whether it was carried out
  if (!runs_of(d, s, d->rank).empty()) {
    SVRCHK(svr_set_present_status(d->ctx, s.status + d->rank));
    SVRCHK(svr_copy_to_swapchain(d->ctx, s.image, d->W, d->H, d->fmt));
    SVRCHK(svr_set_present_status(d->ctx, nullptr));
  }
  HIPCHK(hipEventRecord(s.rendered, d->render));
  if (int e = exchange(d, s)) return e;
  s.in_flight = true;
  d->fifo.push_back(d->cur);
  d->cur = -1;
  return SVR_OK;
}

int svr_dist_wait_frame(SvrDist* d, const void** image_dev) {
  if (!d) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_wait_frame: null argument");
  if (d->fifo.empty()) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_wait_frame: no frame in flight");
  const int slot = d->fifo.front();
  d->fifo.erase(d->fifo.begin());
  SvrDist::Slot& s = d->slots[slot];
  HIPCHK(hipEventSynchronize(s.exchanged));
  // A presented image is a finished one (src/vk_engine.cpp:1226, 1332).  The renderer replays a pass that overflowed
  // its queues — and the present behind it — but only when it next looks (include/svr.h, svr_set_present_status), and
  // it knows nothing of the exchange: the rows that travelled were the slot's old ones.  The status words travelled
  // with them, so every rank sees the same thing here and the repair is collective without a message of its own:
  // fence (the owner replays; the others only wait for their own work), then exchange the slot again.  (A word of 2:
  // the replay ran by itself — any svr_* call may find the overflow — possibly under the exchange: same repair.)
  for (int attempt = 0;; attempt++) {
    bool stale = false;
    for (int r = 0; r < d->world; r++) stale |= s.h_status[r] != 0u;
    if (!stale) break;
    if (attempt == 4) return fail(SVR_ERR_OVERFLOW, "svr_dist_wait_frame: a rank's present stayed void after four exchanges");
    SVRCHK(svr_sync(d->ctx));
    if (s.h_status[d->rank]) HIPCHK(hipMemsetAsync(s.status + d->rank, 0, 4, d->render));  // the replay's present left a 2
    HIPCHK(hipEventRecord(s.rendered, d->render));
    if (int e = exchange(d, s)) return e;
    HIPCHK(hipEventSynchronize(s.exchanged));
    d->replays++;
  }
  // the render stream may reuse the slot once the exchange has read it
  HIPCHK(hipStreamWaitEvent(d->render, s.exchanged, 0));
  s.in_flight = false;
  if (image_dev) *image_dev = s.image;
  return SVR_OK;
}

int svr_dist_read_frame(SvrDist* d, void* dst_host, size_t bytes) {
  if (!d || !dst_host) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_read_frame: null argument");
  const size_t need = (size_t)d->W * d->H * 4;
  if (bytes < need) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_dist_read_frame: buffer too small");
  const void* img = nullptr;
  if (int e = svr_dist_wait_frame(d, &img)) return e;
  HIPCHK(hipMemcpy(dst_host, img, need, hipMemcpyDeviceToHost));
  return SVR_OK;
}

}  // extern "C"
