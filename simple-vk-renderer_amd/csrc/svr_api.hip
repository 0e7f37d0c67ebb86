// svr_api.hip — the C ABI of include/svr.h over the HIP kernels (host code; no kernels here).
//
// Host half of VulkanEngine::draw_geometry (src/vk_engine.cpp:1357-1477): is_visible cull, sort,
// per-draw records (the push constants + bound buffers of the record lambda, :1412-1457) — or, for
// large object counts, handing the objects to k_flatten.hip — then one pass of seven kernels:
//   prologue -> setup -> clip -> count -> offsets -> fill   (internal stream: overlaps the previous tiles)
//   tiles                                                    (caller's stream, after an event)
// The pass is asynchronous like a recorded command buffer; svr_sync / read-backs are the fence.
// Per-pass device buffers only grow.  A pass whose internal queues overflowed writes nothing to the
// targets, and neither does anything after it, until the host has replayed it with larger queues
// ("the operation log" below), so results never depend on the initial capacities.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <string>
#include <vector>

#include "svr_cull.h"
#include "svr_launch.h"

using namespace svr;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess)                                                                              \
      return fail(e_ == hipErrorOutOfMemory ? SVR_ERR_OUT_OF_MEMORY : SVR_ERR_DEVICE,                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                                  \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  int ensure(size_t bytes) {  // contents are NOT preserved
    if (bytes <= cap) return SVR_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 4;
    HIPCHK(hipMalloc(&p, want));
    cap = want;
    return SVR_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct MeshRes {
  SvrVertex* vtx = nullptr;
  uint32_t* idx = nullptr;
  float* groups = nullptr;  // float[6] per 192 indices: box of the vertices they name (setup kernel's chunk culling)
  size_t n_vtx = 0, n_idx = 0;
  bool alive = false;
};
struct ImageRes {
  uint32_t arena_off = 0;   // byte offset of level 0 in the context's texel arena
  size_t bytes = 0;         // all levels
  uint32_t w = 0, h = 0, levels = 0;
  uint32_t lw = 0, lh = 0;  // log2 of the power-of-two padded extent the mip layout is computed from
  uint32_t off[16] = {0};   // = mip_offset(lw, lh, level)
  bool alive = false;
};
struct MaterialRes {
  int pass;
  float cf[4], mr[4];
  uint32_t image, sampler;  // 0-based
};

// software fp32 -> fp16 (RTE) for the one clear colour the host encodes
uint16_t host_f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax >= 0x38800000u) {
    uint32_t mant = ax & 0x7fffffu, h = (((ax >> 23) - 112) << 10) | (mant >> 13), rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  if (ax < 0x33000000u) return (uint16_t)sign;
  uint32_t mant = (ax & 0x7fffffu) | 0x800000u;
  int shift = 126 - (int)(ax >> 23);
  uint32_t h = mant >> shift, rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

}  // namespace

struct SvrContext {
  int device = 0;
  uint32_t W = 0, H = 0;
  int fmt = SVR_COLOR_RGBA16F;
  hipStream_t stream = nullptr;
  void* color_own = nullptr;
  float* depth_own = nullptr;
  void* color = nullptr;
  float* depth = nullptr;
  uint32_t sx = 0, sy = 0, sw = 0, sh = 0;
  uint32_t rstride = 1, roff = 0;      // svr_set_row_interleave
  uint32_t* present_status = nullptr;  // svr_set_present_status

  std::vector<MeshRes> meshes;
  std::vector<ImageRes> images;
  std::vector<SvrSamplerDesc> samplers;
  std::vector<MaterialRes> materials;
  // Texel arena: every image of the context lives in ONE allocation, so a texel's address is a 32-bit byte
  // offset from one wave-uniform base (FrameParams::tex_arena): the fragment stage's eight gathers per pixel
  // are global loads with an SGPR base and a 32-bit VGPR offset instead of 64-bit pointer arithmetic per tap,
  // and records carry 4 bytes per texture, not a pointer.  Grows by reallocation (device copy, after a
  // fence); offsets never change.  Bound: 4 GiB of texels per context.
  uint8_t* tex_arena = nullptr;
  size_t tex_arena_cap = 0, tex_arena_top = 0;
  std::vector<std::pair<size_t, size_t>> tex_holes;  // (offset, bytes) of destroyed images, sorted by offset
  DevBuf tex_table;  // TexBinding[materials + 1]; last slot = scratch binding of svr_draw_tex_image
  size_t tex_slots = 0;
  // resource tables of the device flatten pass (k_flatten.hip), rebuilt when a mesh / material was added
  DevBuf mesh_table, mat_table;
  size_t mesh_table_n = 0, mat_table_n = 0;
  int device_flatten = 0;  // SVR_OPT_DEVICE_FLATTEN: 0 auto (>= 2048 objects), 1 always, 2 never

  // Per-pass device buffers, double-buffered: the geometry+binning stage of pass N+1 runs on the
  // internal stream `gstream` while the tile stage of pass N still reads set N on the caller's
  // stream.  ev_bin: set filled (recorded on gstream); ev_tile: set consumed (the pass's op_done event).
  struct PassSet {
    DevBuf inputs, recs, clipq, bigq, tiles, bins, pairs, flat, sorta;  // flat: keys / triangle counts / chunk bases of k_flatten  // inputs = DrawDesc[] then WaveChunk[] (one H2D copy)
    hipEvent_t ev_bin = nullptr;
    hipEvent_t ev_tile = nullptr;  // not owned: op_done of the pass that used the set last
    bool used = false;
  };
  static const int MAX_OPS = 8;  // operations in flight (log slots)
  static const int NSETS = 4;  // stage 1 of a small pass may run three passes ahead of the tile stage; a large one keeps to one (submit_pass)
  PassSet sets[NSETS];
  int set_pos = 0;
  // operation log (see "the operation log" below)
  struct LoggedOp {
    bool is_pass = false;
    uint32_t seq = 0;  // passes: running number, reported by the device if the pass overflows
    bool timed = false;  // op_start/op_done of the slot bracket the tile kernel: fold into the running mean at retirement
    int slot = 0;  // index into h_counters / op_done
    int fill_kind = 0;  // not a pass: 0 clear, 1 background effect, 2 blit to the swapchain image
    void* clear_rows = nullptr;  // clear: first row, pixel count, format, encoded texel
    uint32_t clear_pixels = 0;
    int clear_fmt = 0;
    uint64_t clear_packed = 0;
    void* target = nullptr;  // background / blit: colour target, its extent, the scissor's rows
    uint32_t tw = 0, th = 0, y_first = 0, n_rows = 0;
    int bg_effect = 0;
    float bg_data[16] = {};
    void* blit_dst = nullptr;
    uint32_t blit_w = 0, blit_h = 0;
    int blit_fmt = 0;
    uint32_t blit_rstride = 1, blit_roff = 0, blit_row_end = 0;  // identity blits of an interleaved pass: its tile rows only
    uint32_t* blit_status = nullptr;
    FrameParams P{};  // pass: parameters as recorded + its draw list
    std::vector<DrawDesc> draws;
    // device-flattened pass: the caller's objects (opaque, then transparent) instead of a draw list
    std::vector<SvrRenderObject> objects;
    uint32_t n_opaque_obj = 0, n_transparent_obj = 0;
  };
  std::deque<LoggedOp> log;
  hipEvent_t op_done[MAX_OPS] = {};
  hipEvent_t op_start[MAX_OPS] = {};  // SVR_OPT_KERNEL_TIMING level 1: start of the slot's tile kernel (rides on its dispatch)
  int op_pos = 0;
  uint32_t replayed = 0;         // passes re-run by recover_from_overflow
  // svr_clear_color deferred into the next pass (the attachment's loadOp CLEAR): see flush_clear
  struct PendingClear {
    bool valid = false;
    void* target = nullptr;
    uint32_t y0 = 0, rows = 0;
    int fmt = 0;
    uint64_t packed = 0;
  } pending_clear;
  uint32_t next_seq = 1;
  uint32_t* h_failed_seq = nullptr;  // pinned; written by the tile kernel of the first failing pass
  uint32_t* d_poison = nullptr;  // sticky device flag: a pass overflowed, later target writes are void
  hipStream_t gstream = nullptr;
  hipStream_t gstream_hi = nullptr;   // the same at the highest priority: stage 1 of small passes (submit_pass)
  hipStream_t last_g = nullptr;       // the one the previous pass used
  hipEvent_t ev_gswitch = nullptr;
  DevBuf d_cvt;
  uint32_t clip_cap = 0, extra_cap = 0, bin_cap = 0;
  uint32_t debug_caps = 0;  // SVR_OPT_QUEUE_CAPS
  // pinned host staging + read-back, one of each per operation-log slot
  void* h_stage[MAX_OPS] = {};  // per log slot
  size_t h_stage_cap[MAX_OPS] = {};
  Counters* h_counters = nullptr;  // pinned, [MAX_OPS]
  uint32_t* h_row_cost = nullptr;  // pinned, [MAX_OPS][ROW_COST_MAX]: tile-row costs posted by every pass's tile kernel
  std::vector<uint32_t> row_cost;  // ... of the pass validated last (svr_get_row_costs), with its scissor rows
  uint32_t row_cost_y0 = 0, row_cost_rows = 0;

  // SVR_OPT_KERNEL_TIMING: ring of event quadruples (before setup, after clip, after fill, after tiles)
  static const int TRING = 16;
  hipEvent_t tev[TRING][5] = {};  // geometry start, after clip, after fill (gstream) | tile start, tile end (stream)
  bool tev_used[TRING] = {};
  bool tev_all[TRING] = {};  // the slot holds all five events (level 2), not just the tile pair
  int tev_pos = 0;
  int kernel_timing = 0;  // 0 off, 1 tile kernel only (two events on the caller's stream), 2 all three stages
  double acc_ms[3] = {0, 0, 0};
  uint32_t acc_n = 0;
  FrameParams last{};        // parameters of the pass enqueued last (debug read-backs)
  bool instrument = false;
  bool tile_cycles = false;
  uint32_t tuning = 0;
  int trace_x = -1, trace_y = -1;
  DevBuf d_trace, d_tile_cycles;
  SvrStats stats{};
};

namespace {

int use_device(SvrContext* ctx) {
  HIPCHK(hipSetDevice(ctx->device));
  return SVR_OK;
}

MeshRes* get_mesh(SvrContext* ctx, SvrMesh h) {
  if (h == 0 || h > ctx->meshes.size() || !ctx->meshes[h - 1].alive) return nullptr;
  return &ctx->meshes[h - 1];
}
ImageRes* get_image(SvrContext* ctx, SvrImage h) {
  if (h == 0 || h > ctx->images.size() || !ctx->images[h - 1].alive) return nullptr;
  return &ctx->images[h - 1];
}

TexBinding make_binding(const ImageRes& im, const SvrSamplerDesc& s) {
  TexBinding tb;
  std::memset(&tb, 0, sizeof(tb));
  uint32_t filters = (uint32_t)s.mag_filter | ((uint32_t)s.min_filter << 1) | ((uint32_t)s.mipmap_mode << 2);
  tb.base_off = im.arena_off;
  tb.wh = im.w | (im.h << 16);
  tb.info = im.lw | (im.lh << 8) | (im.levels << 16) | (filters << 24);
  tb.min_lod = s.min_lod;
  tb.max_lod = s.max_lod;
  return tb;
}

// ---------------------------------------------------------------- texel arena
int finish_pending(SvrContext* ctx);

constexpr size_t ARENA_MAX = (size_t)0xffffff00u;  // offsets are 32-bit
constexpr size_t ARENA_ALIGN = 256;

int arena_alloc(SvrContext* ctx, size_t bytes, uint32_t* off) {
  bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
  for (size_t i = 0; i < ctx->tex_holes.size(); i++) {  // first fit among the holes
    auto& h = ctx->tex_holes[i];
    if (h.second >= bytes) {
      *off = (uint32_t)h.first;
      h.first += bytes;
      h.second -= bytes;
      if (h.second == 0) ctx->tex_holes.erase(ctx->tex_holes.begin() + (long)i);
      return SVR_OK;
    }
  }
  if (ctx->tex_arena_top + bytes > ARENA_MAX)
    return fail(SVR_ERR_OUT_OF_MEMORY, "svr_create_image: more than 4 GiB of texels in one context");
  if (ctx->tex_arena_top + bytes > ctx->tex_arena_cap) {
    // grow: passes in flight read the old arena, so this is a fence; offsets stay valid
    size_t want = std::max(ctx->tex_arena_cap * 2, ctx->tex_arena_top + bytes);
    want = std::min(ARENA_MAX, (want + ((size_t)64 << 20) - 1) & ~(((size_t)64 << 20) - 1));
    if (int e = finish_pending(ctx)) return e;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    uint8_t* grown = nullptr;
    HIPCHK(hipMalloc((void**)&grown, want));
    if (ctx->tex_arena_top) {
      hipError_t r = hipMemcpy(grown, ctx->tex_arena, ctx->tex_arena_top, hipMemcpyDeviceToDevice);
      if (r != hipSuccess) {
        (void)hipFree(grown);
        return fail(SVR_ERR_DEVICE, std::string("hipMemcpy(texel arena): ") + hipGetErrorString(r));
      }
    }
    if (ctx->tex_arena) (void)hipFree(ctx->tex_arena);
    ctx->tex_arena = grown;
    ctx->tex_arena_cap = want;
  }
  *off = (uint32_t)ctx->tex_arena_top;
  ctx->tex_arena_top += bytes;
  return SVR_OK;
}

void arena_free(SvrContext* ctx, size_t off, size_t bytes) {
  bytes = (bytes + ARENA_ALIGN - 1) & ~(ARENA_ALIGN - 1);
  auto& holes = ctx->tex_holes;
  size_t i = 0;
  while (i < holes.size() && holes[i].first < off) i++;
  holes.insert(holes.begin() + (long)i, std::make_pair(off, bytes));
  if (i + 1 < holes.size() && holes[i].first + holes[i].second == holes[i + 1].first) {  // merge with the next
    holes[i].second += holes[i + 1].second;
    holes.erase(holes.begin() + (long)i + 1);
  }
  if (i > 0 && holes[i - 1].first + holes[i - 1].second == holes[i].first) {  // and the previous
    holes[i - 1].second += holes[i].second;
    holes.erase(holes.begin() + (long)i);
  }
  if (!holes.empty() && holes.back().first + holes.back().second == ctx->tex_arena_top) {  // a hole at the top is no hole
    ctx->tex_arena_top = holes.back().first;
    holes.pop_back();
  }
}

// (re)upload the binding table: one slot per material + one scratch slot
int upload_tex_table(SvrContext* ctx, const TexBinding* scratch) {
  size_t n = ctx->materials.size() + 1;
  std::vector<TexBinding> host(n);
  for (size_t i = 0; i < ctx->materials.size(); i++) {
    const MaterialRes& m = ctx->materials[i];
    host[i] = make_binding(ctx->images[m.image], ctx->samplers[m.sampler]);
  }
  if (scratch)
    host[n - 1] = *scratch;
  else
    std::memset(&host[n - 1], 0, sizeof(TexBinding));
  if (ctx->tex_table.cap < n * sizeof(TexBinding)) {
    // the table may be referenced by an in-flight pass: drain before replacing it
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (int e = ctx->tex_table.ensure(n * sizeof(TexBinding) * 2)) return e;
  }
  HIPCHK(hipMemcpyAsync(ctx->tex_table.p, host.data(), n * sizeof(TexBinding), hipMemcpyHostToDevice, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));  // host vector dies here
  ctx->tex_slots = n;
  return SVR_OK;
}

// ---------------------------------------------------------------- pass machinery
// pinned staging buffer of a log slot (free: the slot's previous operation has been retired)
int stage_buffer(SvrContext* ctx, int slot, size_t bytes, void** out) {
  if (ctx->h_stage_cap[slot] < bytes) {
    if (ctx->h_stage[slot]) (void)hipHostFree(ctx->h_stage[slot]);
    ctx->h_stage[slot] = nullptr;
    ctx->h_stage_cap[slot] = 0;
    size_t want = bytes + bytes / 2 + 4096;
    HIPCHK(hipHostMalloc(&ctx->h_stage[slot], want, hipHostMallocDefault));
    ctx->h_stage_cap[slot] = want;
  }
  *out = ctx->h_stage[slot];
  return SVR_OK;
}

// fold one finished slot of the timing ring into the running means
int harvest_timing(SvrContext* ctx, int slot) {
  if (!ctx->tev_used[slot]) return SVR_OK;
  HIPCHK(hipEventSynchronize(ctx->tev[slot][4]));
  const int from[3] = {0, 1, 3}, to[3] = {1, 2, 4};
  for (int k = ctx->tev_all[slot] ? 0 : 2; k < 3; k++) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, ctx->tev[slot][from[k]], ctx->tev[slot][to[k]]));
    ctx->acc_ms[k] += ms;
  }
  ctx->acc_n++;
  ctx->tev_used[slot] = false;
  return SVR_OK;
}

// head of a set's tile buffer: Counters (96 B) + 80 class counters (FrameParams::cls_count) + ROW_COST_MAX row costs, then tile_count
constexpr size_t TILE_HEAD_BYTES = sizeof(Counters) + (80 + ROW_COST_MAX) * sizeof(uint32_t);
static_assert(TILE_HEAD_BYTES % 16 == 0, "tile buffer head layout");

// size the per-pass buffers for P.n_tris and the current capacities, fill the pointers
int bind_pass_buffers(SvrContext* ctx, FrameParams& P, int set_index) {
  SvrContext::PassSet& set = ctx->sets[set_index];
  if (int e = set.recs.ensure(((size_t)P.n_tris + ctx->extra_cap) * sizeof(TriRec))) return e;
  if (int e = set.clipq.ensure((size_t)ctx->clip_cap * sizeof(ClipItem))) return e;
  if (int e = set.bigq.ensure(((size_t)P.n_tris + 64) * sizeof(uint32_t))) return e;
  if (int e = set.tiles.ensure(TILE_HEAD_BYTES + ((size_t)P.n_tiles * 13 + 8 + (size_t)SPLIT_EXTRA * 8) * sizeof(uint32_t))) return e;
  if (int e = set.bins.ensure((size_t)ctx->bin_cap * sizeof(uint32_t))) return e;
  if (int e = set.pairs.ensure((size_t)ctx->bin_cap * 12)) return e;
  if (int e = set.sorta.ensure((size_t)ctx->bin_cap * 2 * sizeof(unsigned long long))) return e;
  P.recs = (TriRec*)set.recs.p;
  P.extra_cap = ctx->extra_cap;
  P.clip_queue = (ClipItem*)set.clipq.p;
  P.clip_cap = ctx->clip_cap;
  P.big_queue = (uint32_t*)set.bigq.p;
  P.counters = (Counters*)set.tiles.p;
  P.cls_count = (uint32_t*)((char*)set.tiles.p + sizeof(Counters));
  P.row_cost = P.cls_count + 80;
  P.tile_count = (uint32_t*)((char*)set.tiles.p + TILE_HEAD_BYTES);
  P.tile_offset = P.tile_count + (((size_t)P.n_tiles * 2 + 3) & ~(size_t)3);  // 16-byte aligned
  P.tile_info = (uint4*)(P.tile_offset + (((size_t)P.n_tiles * 2 + 3) & ~(size_t)3));  // 8 words per tile
  P.tile_order = (uint32_t*)P.tile_info + ((size_t)P.n_tiles + SPLIT_EXTRA) * 8;  // the quarters of split tiles head tile_info
  P.pairs = (uint2*)set.pairs.p;
  P.pair_slot = (uint32_t*)((char*)set.pairs.p + (size_t)ctx->bin_cap * 8);
  P.bins = (uint32_t*)set.bins.p;
  P.bin_cap = ctx->bin_cap;
  P.sort_arena = (unsigned long long*)set.sorta.p;
  P.sort_cap = ctx->bin_cap * 2u;  // a sorted bin needs at most twice its entries (power-of-two padding)
  P.poison = ctx->d_poison;
  P.host_failed_seq = ctx->h_failed_seq;
  return SVR_OK;
}

// Enqueue one pass.  Stage 1 (gstream): prologue (inputs + zeroing), setup, clip, bin count, offsets, bin fill
// -> ev_bin.  Stage 2 (caller's stream): wait ev_bin, tile kernel, counters to the host
// (report_kernel) -> op_done.  The caller sees stream order (everything it enqueued before the call precedes the tile
// stage, the only one that touches the targets); stage 1 depends on host inputs alone, so it overlaps
// the tile stages of the passes before it.
int submit_pass(SvrContext* ctx, FrameParams P, const std::vector<DrawDesc>& draws, int op_slot, uint32_t seq, bool pipe,
                const SvrContext::LoggedOp* flat_op = nullptr) {
  // queue capacities: generous first guesses; overflow -> replay (recover_from_overflow)
  if (ctx->debug_caps) {  // SVR_OPT_QUEUE_CAPS: start tiny so that tests reach the replay path
    ctx->clip_cap = std::max<uint32_t>(ctx->clip_cap, ctx->debug_caps);
    ctx->extra_cap = std::max<uint32_t>(ctx->extra_cap, ctx->debug_caps);
    ctx->bin_cap = std::max<uint32_t>(ctx->bin_cap, ctx->debug_caps);
  } else {
    ctx->clip_cap = std::max<uint32_t>(ctx->clip_cap, std::max<uint32_t>(65536u, P.n_tris / 4u));
    ctx->extra_cap = std::max<uint32_t>(ctx->extra_cap, std::max<uint32_t>(65536u, P.n_tris / 2u));
    ctx->bin_cap = std::max<uint32_t>(ctx->bin_cap, std::max<uint32_t>(1u << 22, P.n_tris * 8u));
  }
  const int set_index = ctx->set_pos;
  ctx->set_pos = (ctx->set_pos + 1) % SvrContext::NSETS;
  SvrContext::PassSet& set = ctx->sets[set_index];
  // Stage 1 of a pass of few tiles (a 1920x1080 frame, a band of a sharded one) runs at the highest stream priority:
  // such a pass is bounded by stage 1 — its setup kernel finds the CUs taken by the tile kernel's first, longest
  // workgroups (21 us alone, 53 us beside it) — and with priority its workgroups get the slots that come free
  // (1080p: -5 % per frame).  A 4K frame is bounded by its tile kernel and loses 0.8 % to the same favour.
  // Either stream is made when a pass first needs it: the runtime maps a process's streams onto a handful of hardware
  // queues (four by default), and streams that share one serialise — a context that only ever renders one size of pass
  // must not take a queue it never uses (two contexts with both streams in one process: stage 1 and the tile kernel of
  // the second ended up in ONE queue, 0.093 -> 0.27 ms per 1080p frame).
  hipStream_t s = ctx->stream, g = ctx->stream;
  if (pipe) {
#ifdef SVR_AB_STAGE1_HI  // A/B builds only: stage 1 of every pass on the high-priority stream
    const bool hi = true;
#else
    const bool hi = P.n_tiles <= SPLIT_TILES_MAX;
#endif
    hipStream_t& slot = hi ? ctx->gstream_hi : ctx->gstream;
    if (!slot) {
      if (hi) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&slot, hipStreamNonBlocking, greatest) != hipSuccess) {
          (void)hipGetLastError();  // no priorities here: an ordinary stream does the job, a little later
          slot = nullptr;
          HIPCHK(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
        }
      } else {
        HIPCHK(hipStreamCreateWithFlags(&slot, hipStreamNonBlocking));
      }
    }
    g = slot;
  }
  if (pipe) {
    if (ctx->last_g && ctx->last_g != g) {  // keep stage 1 of consecutive passes in order across the two streams
      HIPCHK(hipEventRecord(ctx->ev_gswitch, ctx->last_g));
      HIPCHK(hipStreamWaitEvent(g, ctx->ev_gswitch, 0));
    }
    ctx->last_g = g;
  }
  // per-pass inputs: draws + chunks through pinned staging, one copy
  const size_t n_objects = flat_op ? flat_op->objects.size() : 0;
  size_t draw_bytes = (flat_op ? n_objects : draws.size()) * sizeof(DrawDesc), chunk_bytes = (size_t)P.n_chunks * sizeof(WaveChunk);
  if (int e = set.inputs.ensure(std::max<size_t>(draw_bytes + chunk_bytes + 16, 256))) return e;
  if (flat_op)
    if (int e = set.flat.ensure(n_objects * (16 + sizeof(SvrRenderObject)) + 128)) return e;
  if (int e = bind_pass_buffers(ctx, P, set_index)) return e;
  // How far stage 1 runs ahead.  A pass of few tiles (a band of a sharded frame: stage 1 56 us, tiles 50 us) is bounded
  // by stage 1, which then wants to run back to back: it only waits for its set, last read by the tile stage of
  // NSETS passes ago (a band of an eight-way split: -16 % per frame against two sets, with the priority above).  A 4K
  // frame is bounded by its tile kernel, and stage-1 kernels that arrive earlier only take CU time from it (+0.6 %):
  // it waits for the tile stage of two passes back (which is behind that of NSETS passes ago in the stream).
  if (pipe) {
    SvrContext::PassSet& gate = P.n_tiles > SPLIT_TILES_MAX ? ctx->sets[(set_index + SvrContext::NSETS - 2) % SvrContext::NSETS] : set;
    if (gate.used) HIPCHK(hipStreamWaitEvent(g, gate.ev_tile, 0));
    else if (set.used) HIPCHK(hipStreamWaitEvent(g, set.ev_tile, 0));
  }
  void* stage = nullptr;
  if (int e = stage_buffer(ctx, op_slot, (flat_op ? n_objects * sizeof(SvrRenderObject) : draw_bytes + chunk_bytes) + 64, &stage)) return e;
  P.host_counters = &ctx->h_counters[op_slot];
  P.host_row_cost = ctx->h_row_cost + (size_t)op_slot * ROW_COST_MAX;
  P.op_seq = seq;
  if (flat_op) {  // the objects themselves are the input; cull, sort, draw records and chunks happen on the device
    std::memcpy(stage, flat_op->objects.data(), n_objects * sizeof(SvrRenderObject));
  } else {
    std::memcpy(stage, draws.data(), draw_bytes);
    WaveChunk* ch = reinterpret_cast<WaveChunk*>((char*)stage + draw_bytes);
    size_t ci = 0;
    for (size_t di = 0; di < draws.size(); di++)
      for (uint32_t k = 0, nk = chunk_count(draws[di].first_index, draws[di].tri_count); k < nk; k++) {
        ch[ci].draw = (uint32_t)di;
        ch[ci].first_tri = chunk_first(draws[di].first_index, k);
        ci++;
      }
  }
  P.draws = (const DrawDesc*)set.inputs.p;
  P.chunks = (const WaveChunk*)((const char*)set.inputs.p + draw_bytes);  // DrawDesc is 192 B: stays 16-byte aligned

  int ts = -1;
  if (ctx->kernel_timing >= 2) {
    ts = ctx->tev_pos;
    ctx->tev_pos = (ctx->tev_pos + 1) % SvrContext::TRING;
    if (int e = harvest_timing(ctx, ts)) return e;
    for (int k = 0; k < 5; k++)
      if (!ctx->tev[ts][k]) HIPCHK(hipEventCreate(&ctx->tev[ts][k]));
  }
  // inputs out of the staging buffer + zero the counters, class counters and tile_count (adjacent)
  launch_prologue(stage, set.inputs.p, flat_op ? 0 : draw_bytes + chunk_bytes, P.counters,
                  TILE_HEAD_BYTES + (size_t)P.n_tiles * 2 * sizeof(uint32_t), flat_op ? 0u : (uint32_t)draws.size(), P.scene, g);
  if (flat_op) {
    FlattenParams F;
    std::memset(&F, 0, sizeof(F));
    F.objects = (const SvrRenderObject*)stage;
    F.n_opaque = flat_op->n_opaque_obj;
    F.n_transparent = flat_op->n_transparent_obj;
    std::memcpy(F.viewproj, P.scene.viewproj, 64);
    F.meshes = (const MeshEntry*)ctx->mesh_table.p;
    F.materials = (const MatEntry*)ctx->mat_table.p;
    F.keys = (unsigned long long*)set.flat.p;
    F.draw_tris = (uint32_t*)((char*)set.flat.p + n_objects * 8);
    F.chunk_base = F.draw_tris + n_objects;
    F.objects_dev = (SvrRenderObject*)((char*)set.flat.p + ((n_objects * 16 + 63) & ~(size_t)63));
    F.draws = (DrawDesc*)set.inputs.p;
    F.chunks = (WaveChunk*)((char*)set.inputs.p + draw_bytes);
    F.counters = P.counters;
    launch_flatten(F, g);
  }
  const bool all_stages = ctx->kernel_timing >= 2;
  if (ts >= 0 && all_stages) HIPCHK(hipEventRecord(ctx->tev[ts][0], g));
  launch_setup(P, g);
  if (ts >= 0 && all_stages) HIPCHK(hipEventRecord(ctx->tev[ts][1], g));
  launch_bin_count(P, g);
  launch_bin_scan(P, g);
  launch_bin_fill(P, g, pipe ? set.ev_bin : nullptr);  // ev_bin rides on the fill kernel's dispatch
  if (ts >= 0 && all_stages) HIPCHK(hipEventRecord(ctx->tev[ts][2], g));
  if (pipe) HIPCHK(hipStreamWaitEvent(s, set.ev_bin, 0));
  if (ts >= 0) HIPCHK(hipEventRecord(ctx->tev[ts][3], s));
  // op_done rides on the pass's last kernel; with kernel timing level 1 op_start rides on the tile kernel too
  launch_tiles(P, ctx->fmt, P.instrument != 0, s, ctx->kernel_timing == 1 ? ctx->op_start[op_slot] : nullptr, ctx->op_done[op_slot]);
  if (ts >= 0) {
    HIPCHK(hipEventRecord(ctx->tev[ts][4], s));
    ctx->tev_used[ts] = true;
    ctx->tev_all[ts] = all_stages;
  }
  HIPCHK(hipGetLastError());
  // the one event of the pass: its counters are on the host, its set and staging buffer are free
  set.ev_tile = ctx->op_done[op_slot];
  set.used = true;
  ctx->last = P;
  return SVR_OK;
}

void note_pass_stats(SvrContext* ctx, const FrameParams&, const Counters& c) {  // instrumented passes only
  ctx->stats.bin_entries = c.total_entries;
  ctx->stats.rasterized_fragments = c.rasterized;
  ctx->stats.shaded_fragments = c.shaded;
  ctx->stats.binned_triangles = c.binned;
}

// the tile rows' costs of a finished pass (posted by its tile kernel before anything else): svr_get_row_costs
void note_row_costs(SvrContext* ctx, const FrameParams& Pk, int slot) {
  const uint32_t* src = ctx->h_row_cost + (size_t)slot * ROW_COST_MAX;
  ctx->row_cost.assign(src, src + std::min<uint32_t>(Pk.tiles_y, ROW_COST_MAX));
  ctx->row_cost_y0 = Pk.sy;
  ctx->row_cost_rows = Pk.sh;
}

void note_flatten_stats(SvrContext* ctx, const Counters& c) {  // device-flattened passes learn these late
  ctx->stats.drawcall_count = (int32_t)c.flat_draws;
  ctx->stats.triangle_count = (int32_t)c.flat_tris;
  ctx->stats.culled_draws = c.flat_culled;
}

// ---------------------------------------------------------------- the operation log
// Passes run asynchronously and several deep, so the host learns of a queue overflow late.  The
// guarantee "results never depend on queue capacities" is kept like this: the tile kernel of a pass
// that overflowed writes nothing and raises a sticky device flag (poison); every later target-writing
// kernel of this context (tile kernels, clears) sees the flag and writes nothing either, so the
// targets freeze in the state before the failed pass.  The host keeps every target-writing operation
// in a log until its completion event has fired and its counters were checked; on an overflow it
// drains the device, lowers the flag, grows the queues and replays the log from the failed operation
// on, in order.  (Work the caller itself enqueues between passes is not in the log; SvrStats.
// replayed_passes tells such a caller that a replay happened — see dist.py.)
int log_slot(SvrContext* ctx, int* slot);
int retire_ops(SvrContext* ctx, bool blocking);
int flush_clear(SvrContext* ctx);

// replaying: the operation is run again by recover_from_overflow (a present then reports 2 instead of 0)
int submit_clear(SvrContext* ctx, const SvrContext::LoggedOp& op, bool replaying = false) {  // every logged operation that is not a pass
  if (op.fill_kind == 0)
    launch_fill_color(op.clear_rows, op.clear_pixels, op.clear_fmt, op.clear_packed, ctx->d_poison, ctx->stream);
  else if (op.fill_kind == 1)
    launch_background(op.target, op.clear_fmt, op.tw, op.th, op.y_first, op.n_rows, op.bg_effect, op.bg_data, ctx->d_poison, ctx->stream);
  else
    launch_blit(op.target, op.clear_fmt, op.tw, op.th, op.blit_dst, op.blit_w, op.blit_h, op.y_first, op.n_rows, op.blit_fmt, ctx->d_poison,
                op.blit_rstride, op.blit_roff, op.blit_row_end, op.blit_status, replaying ? 2u : 0u, ctx->stream);
  HIPCHK(hipGetLastError());
  return SVR_OK;
}

int recover_from_overflow(SvrContext* ctx) {
  HIPCHK(hipStreamSynchronize(ctx->stream));
  if (ctx->gstream) HIPCHK(hipStreamSynchronize(ctx->gstream));
  if (ctx->gstream_hi) HIPCHK(hipStreamSynchronize(ctx->gstream_hi));
  const uint32_t failed_seq = *(volatile uint32_t*)ctx->h_failed_seq;
  *ctx->h_failed_seq = 0;
  for (SvrContext::LoggedOp& op : ctx->log) {
    if (!op.is_pass) {
      HIPCHK(hipMemsetAsync(ctx->d_poison, 0, sizeof(uint32_t), ctx->stream));
      if (int e = submit_clear(ctx, op, true)) return e;
      continue;
    }
    bool done = false;
    Counters c;
    std::memset(&c, 0, sizeof(c));
    // the pass that failed reported its counters with its number (tile_kernel): grow before the first replay.
    // The passes behind it were only void, not known to overflow: they start from the capacities as they are.
    if (op.seq == failed_seq && ctx->h_counters[op.slot].overflow) c = ctx->h_counters[op.slot];
    for (int attempt = 0; attempt < 13 && !done; attempt++) {
      if (c.overflow & 1u) ctx->clip_cap = std::max<uint32_t>(ctx->clip_cap * 2u, c.n_clip + 1024u);
      if (c.overflow & 2u) ctx->extra_cap = std::max<uint32_t>(ctx->extra_cap * 2u, c.n_extra + 1024u);
      if (c.overflow & 4u) {
        uint32_t need = std::max(c.total_entries, c.n_pairs + c.n_pairs_rest);
        ctx->bin_cap = std::max<uint32_t>(ctx->bin_cap * 2u, need + need / 4u);
      }
      HIPCHK(hipMemsetAsync(ctx->d_poison, 0, sizeof(uint32_t), ctx->stream));
      if (int e = submit_pass(ctx, op.P, op.draws, op.slot, op.seq, false, op.P.flatten ? &op : nullptr)) return e;
      HIPCHK(hipStreamSynchronize(ctx->stream));
      HIPCHK(hipMemcpy(&c, ctx->last.counters, sizeof(Counters), hipMemcpyDeviceToHost));
      *ctx->h_failed_seq = 0;
      done = c.overflow == 0;
    }
    if (!done) {
      ctx->log.clear();
      return fail(SVR_ERR_OVERFLOW, "a pass kept overflowing its internal queues after 12 replays");
    }
    if (op.P.instrument) note_pass_stats(ctx, op.P, c);
    if (op.P.instrument && c.hiz_bad) return fail(SVR_ERR_DEVICE, "internal check failed: the hierarchical depth test dropped a fragment that wins (" + std::to_string(c.hiz_bad) + ")");
    if (op.P.flatten) note_flatten_stats(ctx, c);
    note_row_costs(ctx, op.P, op.slot);
    ctx->replayed++;
  }
  HIPCHK(hipMemsetAsync(ctx->d_poison, 0, sizeof(uint32_t), ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  ctx->log.clear();
  return SVR_OK;
}

// validate finished operations front to back; blocking = wait for all of them (a fence).
// Clears record no event of their own (an event between two kernels is a bubble in the stream): a
// clear is done when a pass behind it is, or when the stream has drained.
int retire_ops(SvrContext* ctx, bool blocking) {
  while (!ctx->log.empty()) {
    size_t k = 0;  // first pass at or behind the front
    while (k < ctx->log.size() && !ctx->log[k].is_pass) k++;
    if (k == ctx->log.size()) {  // only clears left
      if (!blocking) return SVR_OK;
      HIPCHK(hipStreamSynchronize(ctx->stream));
      ctx->log.clear();
      return SVR_OK;
    }
    const int slot = ctx->log[k].slot;
    if (blocking) {
      HIPCHK(hipEventSynchronize(ctx->op_done[slot]));
    } else {
      hipError_t q = hipEventQuery(ctx->op_done[slot]);
      if (q == hipErrorNotReady) return SVR_OK;
      HIPCHK(q);
    }
    // the device names the first pass that overflowed (tile_kernel); everything from it on is void
    const uint32_t failed = *(volatile uint32_t*)ctx->h_failed_seq;
    if (failed != 0 && failed == ctx->log[k].seq) {
      // the clears in front of the failed pass did land: only it and what follows is replayed
      ctx->log.erase(ctx->log.begin(), ctx->log.begin() + (long)k);
      return recover_from_overflow(ctx);
    }
    if (ctx->log[k].timed) {  // both events belong to the tile kernel's dispatch: its duration, no extra packets
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ctx->op_start[slot], ctx->op_done[slot]) == hipSuccess) {
        ctx->acc_ms[2] += ms;
        ctx->acc_n++;
      }
    }
    if (ctx->log[k].P.instrument) note_pass_stats(ctx, ctx->log[k].P, ctx->h_counters[slot]);
    if (ctx->log[k].P.instrument && ctx->h_counters[slot].hiz_bad) {
      const uint32_t bad = ctx->h_counters[slot].hiz_bad;
      ctx->log.erase(ctx->log.begin(), ctx->log.begin() + (long)k + 1);
      return fail(SVR_ERR_DEVICE, "internal check failed: the hierarchical depth test dropped a fragment that wins (" + std::to_string(bad) + ")");
    }
    if (ctx->log[k].P.flatten) note_flatten_stats(ctx, ctx->h_counters[slot]);
    note_row_costs(ctx, ctx->log[k].P, slot);
    ctx->log.erase(ctx->log.begin(), ctx->log.begin() + (long)k + 1);
  }
  return SVR_OK;
}

// a free slot of the counters/event/staging ring for the next logged operation (waits if full)
int log_slot(SvrContext* ctx, int* slot) {
  if ((int)ctx->log.size() >= SvrContext::MAX_OPS) {
    if (int e = retire_ops(ctx, false)) return e;
    if ((int)ctx->log.size() >= SvrContext::MAX_OPS) {
      bool any_pass = false;
      for (const SvrContext::LoggedOp& op : ctx->log) any_pass |= op.is_pass;
      if (any_pass) {
        for (const SvrContext::LoggedOp& op : ctx->log)
          if (op.is_pass) {
            HIPCHK(hipEventSynchronize(ctx->op_done[op.slot]));
            break;
          }
        if (int e = retire_ops(ctx, false)) return e;
      } else if (int e = retire_ops(ctx, true)) {
        return e;
      }
    }
  }
  *slot = ctx->op_pos;
  ctx->op_pos = (ctx->op_pos + 1) % SvrContext::MAX_OPS;
  return SVR_OK;
}

// A clear of whole scissor rows is not run when it is asked for: the pass that follows writes every
// pixel of those rows anyway (its tile grid covers the scissor), so it takes the clear value for the
// pixels it does not cover and the separate 8-bytes-per-pixel fill disappears — what a Vulkan renderer
// gets from loadOp = CLEAR instead of a clear command.  Anything else that touches or exposes the
// target first (another operation, a read-back, a fence, a change of targets) runs the clear as its
// own kernel here.  SVR_OPT_TUNING bit 2 turns the deferral off.
int flush_clear(SvrContext* ctx) {
  if (!ctx->pending_clear.valid) return SVR_OK;
  const SvrContext::PendingClear pc = ctx->pending_clear;
  ctx->pending_clear.valid = false;
  size_t px_bytes = pc.fmt == SVR_COLOR_RGBA16F ? 8 : 4;
  int slot = 0;
  if (int e = log_slot(ctx, &slot)) return e;
  ctx->log.emplace_back();
  SvrContext::LoggedOp& op = ctx->log.back();
  op.slot = slot;
  op.clear_rows = (char*)pc.target + (size_t)pc.y0 * ctx->W * px_bytes;
  op.clear_pixels = ctx->W * pc.rows;
  op.clear_fmt = pc.fmt;
  op.clear_packed = pc.packed;
  return submit_clear(ctx, op);
}

int finish_pending(SvrContext* ctx) {  // the fence
  if (int e = flush_clear(ctx)) return e;
  return retire_ops(ctx, true);
}
int poll_pending(SvrContext* ctx) { return (ctx->tuning & TUNE_NO_POLL) ? SVR_OK : retire_ops(ctx, false); }

// the scissor's 32-row tile rows this context renders (svr_set_row_interleave: index % rstride == roff)
uint32_t owned_tile_rows(const SvrContext* ctx) {
  const uint32_t all = (ctx->sh + TILE - 1) / TILE;
  return all > ctx->roff ? (all - ctx->roff + ctx->rstride - 1) / ctx->rstride : 0u;
}

int fill_frame_params(SvrContext* ctx, const SvrSceneData* scene, uint64_t n_tris64, size_t n_chunks, FrameParams& P) {
  if (n_tris64 >= 0x3ffffff0ull) return fail(SVR_ERR_UNSUPPORTED, "more than 2^30 triangles in one pass");
  std::memset(&P, 0, sizeof(P));
  P.color = ctx->color;
  P.depth = ctx->depth;
  P.W = ctx->W;
  P.H = ctx->H;
  P.sx = ctx->sx;
  P.sy = ctx->sy;
  P.sw = ctx->sw;
  P.sh = ctx->sh;
  P.tiles_x = (ctx->sw + TILE - 1) / TILE;
  P.rstride = ctx->rstride;
  P.roff = ctx->roff;
  P.tiles_y = owned_tile_rows(ctx);
  P.n_tiles = P.tiles_x * P.tiles_y;
  P.n_tris = (uint32_t)n_tris64;
  P.n_chunks = (uint32_t)n_chunks;
  P.tex = (const TexBinding*)ctx->tex_table.p;
  P.tex_arena = ctx->tex_arena;
  P.instrument = ctx->instrument ? 1u : 0u;
  P.trace_x = ctx->trace_x;
  P.trace_y = ctx->trace_y;
  P.trace_buf = (ctx->instrument && ctx->trace_x >= 0) ? (float*)ctx->d_trace.p : nullptr;
  P.tuning = ctx->tuning;
#ifndef SVR_AB_SPLIT_ALL  // A/B builds only (tools/build_variant.sh): the quarter path for passes of any size
  if (P.n_tiles > SPLIT_TILES_MAX) P.tuning |= TUNE_NO_SPLIT;  // svr_device.h: no tile of such a pass is worth splitting
#endif
  P.tile_cycles = nullptr;
  if (ctx->tile_cycles) {
    if (int e = ctx->d_tile_cycles.ensure((size_t)P.n_tiles * 16)) return e;
    P.tile_cycles = (uint32_t*)ctx->d_tile_cycles.p;
  }
  if (scene) P.scene = *scene;
  // a deferred clear of exactly the rows this pass covers rides along; any other one runs now
  const SvrContext::PendingClear& pc = ctx->pending_clear;
  if (pc.valid && pc.target == ctx->color && pc.fmt == ctx->fmt && pc.y0 == ctx->sy && pc.rows == ctx->sh && ctx->sx == 0 &&
      ctx->sw == ctx->W) {
    P.lazy_clear = 1u;
    P.clear_lo = (uint32_t)pc.packed;
    P.clear_hi = (uint32_t)(pc.packed >> 32);
    ctx->pending_clear.valid = false;
  } else if (int e = flush_clear(ctx)) {
    return e;
  }
  return SVR_OK;
}

int run_pass(SvrContext* ctx, const SvrSceneData* scene, std::vector<DrawDesc>& draws) {
  if (int e = poll_pending(ctx)) return e;
  // sequence numbers + wave chunks
  uint64_t n_tris64 = 0;
  size_t n_chunks = 0;
  for (DrawDesc& d : draws) {
    d.tri_base = (uint32_t)n_tris64;
    n_tris64 += d.tri_count;
    n_chunks += chunk_count(d.first_index, d.tri_count);
  }
  FrameParams P;
  const SvrContext::PendingClear asked = ctx->pending_clear;  // folded into P.lazy_clear below: put back if the pass is not enqueued
  if (int e = fill_frame_params(ctx, scene, n_tris64, n_chunks, P)) return e;
  int slot = 0;
  if (int e = log_slot(ctx, &slot)) {
    if (P.lazy_clear) ctx->pending_clear = asked;
    return e;
  }
  ctx->log.emplace_back();
  SvrContext::LoggedOp& op = ctx->log.back();
  op.is_pass = true;
  op.seq = ctx->next_seq++;
  if (ctx->next_seq == 0) ctx->next_seq = 1;
  op.slot = slot;
  op.timed = ctx->kernel_timing == 1;
  op.P = P;
  op.draws.swap(draws);
  std::memset(&ctx->h_counters[slot], 0, sizeof(Counters));
  if (int e = submit_pass(ctx, op.P, op.draws, slot, op.seq, !(ctx->tuning & TUNE_NO_PIPELINE))) {
    ctx->log.pop_back();
    if (P.lazy_clear) ctx->pending_clear = asked;
    return e;
  }
  return SVR_OK;
}

// (re)upload the handle -> resource tables the device flatten pass reads
int upload_flatten_tables(SvrContext* ctx) {
  if (ctx->mesh_table_n == ctx->meshes.size() && ctx->mat_table_n == ctx->materials.size()) return SVR_OK;
  if (int e = finish_pending(ctx)) return e;  // a pass in flight may be reading the old tables
  std::vector<MeshEntry> me(ctx->meshes.size());
  for (size_t i = 0; i < me.size(); i++) {
    me[i].vtx = ctx->meshes[i].vtx;
    me[i].idx = ctx->meshes[i].idx;
    me[i].groups = ctx->meshes[i].groups;
    me[i].pad = 0;
  }
  std::vector<MatEntry> ma(ctx->materials.size());
  for (size_t i = 0; i < ma.size(); i++) {
    std::memset(&ma[i], 0, sizeof(MatEntry));
    std::memcpy(ma[i].cf, ctx->materials[i].cf, 16);
    ma[i].pass = (uint32_t)ctx->materials[i].pass;
  }
  if (int e = ctx->mesh_table.ensure(std::max<size_t>(me.size() * sizeof(MeshEntry), 256))) return e;
  if (int e = ctx->mat_table.ensure(std::max<size_t>(ma.size() * sizeof(MatEntry), 256))) return e;
  if (!me.empty()) HIPCHK(hipMemcpy(ctx->mesh_table.p, me.data(), me.size() * sizeof(MeshEntry), hipMemcpyHostToDevice));
  if (!ma.empty()) HIPCHK(hipMemcpy(ctx->mat_table.p, ma.data(), ma.size() * sizeof(MatEntry), hipMemcpyHostToDevice));
  ctx->mesh_table_n = me.size();
  ctx->mat_table_n = ma.size();
  return SVR_OK;
}

// draw_geometry with cull, sort and the draw records left to the device (k_flatten.hip): the host only
// validates, sums the upper bounds that size buffers and grids, and hands the objects over.
int run_pass_flatten(SvrContext* ctx, const SvrSceneData* scene, const SvrRenderObject* opaque, size_t n_opaque,
                     const SvrRenderObject* transparent, size_t n_transparent, uint64_t tris_max, size_t chunks_max) {
  if (int e = poll_pending(ctx)) return e;
  if (int e = upload_flatten_tables(ctx)) return e;
  FrameParams P;
  const SvrContext::PendingClear asked = ctx->pending_clear;
  if (int e = fill_frame_params(ctx, scene, tris_max, chunks_max, P)) return e;
  P.flatten = 1u;
  int slot = 0;
  if (int e = log_slot(ctx, &slot)) {
    if (P.lazy_clear) ctx->pending_clear = asked;
    return e;
  }
  ctx->log.emplace_back();
  SvrContext::LoggedOp& op = ctx->log.back();
  op.is_pass = true;
  op.seq = ctx->next_seq++;
  if (ctx->next_seq == 0) ctx->next_seq = 1;
  op.slot = slot;
  op.timed = ctx->kernel_timing == 1;
  op.P = P;
  op.objects.reserve(n_opaque + n_transparent);
  op.objects.insert(op.objects.end(), opaque, opaque + n_opaque);
  op.objects.insert(op.objects.end(), transparent, transparent + n_transparent);
  op.n_opaque_obj = (uint32_t)n_opaque;
  op.n_transparent_obj = (uint32_t)n_transparent;
  std::memset(&ctx->h_counters[slot], 0, sizeof(Counters));
  if (int e = submit_pass(ctx, op.P, op.draws, slot, op.seq, !(ctx->tuning & TUNE_NO_PIPELINE), &op)) {
    ctx->log.pop_back();
    if (P.lazy_clear) ctx->pending_clear = asked;
    return e;
  }
  return SVR_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* svr_last_error(void) { return g_err.c_str(); }
const char* svr_backend_name(void) { return "hip-gfx950"; }

int svr_create(const SvrConfig* cfg, SvrContext** out) {
  if (!cfg || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: null argument");
  if (cfg->width == 0 || cfg->height == 0 || cfg->width > 16384 || cfg->height > 16384)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: extent must be in 1..16384");
  if (cfg->color_format != SVR_COLOR_RGBA16F && cfg->color_format != SVR_COLOR_RGBA8)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: unknown colour format");
  int n_dev = 0;
  hipError_t e = hipGetDeviceCount(&n_dev);
  if (e != hipSuccess || n_dev <= 0)
    return fail(SVR_ERR_DEVICE, std::string("svr_create: no HIP device (") + hipGetErrorString(e) +
                                    "); this library has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= n_dev) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: bad device ordinal");
  HIPCHK(hipSetDevice(cfg->device));
  SvrContext* ctx = new SvrContext();
  ctx->device = cfg->device;
  ctx->W = cfg->width;
  ctx->H = cfg->height;
  ctx->fmt = cfg->color_format;
  ctx->sw = ctx->W;
  ctx->sh = ctx->H;
  size_t n = (size_t)ctx->W * ctx->H;
  size_t cbytes = n * (ctx->fmt == SVR_COLOR_RGBA16F ? 8 : 4);
  auto bail = [&](hipError_t err, const char* what) {
    std::string msg = std::string(what) + ": " + hipGetErrorString(err);
    svr_destroy(ctx);
    return fail(err == hipErrorOutOfMemory ? SVR_ERR_OUT_OF_MEMORY : SVR_ERR_DEVICE, msg);
  };
  hipError_t r;
  if ((r = hipMalloc(&ctx->color_own, cbytes)) != hipSuccess) return bail(r, "hipMalloc(color)");
  if ((r = hipMalloc((void**)&ctx->depth_own, n * 4)) != hipSuccess) return bail(r, "hipMalloc(depth)");
  if ((r = hipMemset(ctx->color_own, 0, cbytes)) != hipSuccess) return bail(r, "hipMemset(color)");
  if ((r = hipMemset(ctx->depth_own, 0, n * 4)) != hipSuccess) return bail(r, "hipMemset(depth)");
  ctx->color = ctx->color_own;
  ctx->depth = ctx->depth_own;
  // (the internal streams are made by the first pass that needs them: submit_pass)
  if ((r = hipEventCreateWithFlags(&ctx->ev_gswitch, hipEventDisableTiming)) != hipSuccess) return bail(r, "hipEventCreate");
  for (int i = 0; i < SvrContext::NSETS; i++) {
    if ((r = hipEventCreateWithFlags(&ctx->sets[i].ev_bin, hipEventDisableTiming)) != hipSuccess) return bail(r, "hipEventCreate");
  }
  if ((r = hipHostMalloc((void**)&ctx->h_counters, sizeof(Counters) * SvrContext::MAX_OPS, hipHostMallocDefault)) != hipSuccess)
    return bail(r, "hipHostMalloc");
  for (int i = 0; i < SvrContext::MAX_OPS; i++)
    if ((r = hipEventCreate(&ctx->op_done[i])) != hipSuccess || (r = hipEventCreate(&ctx->op_start[i])) != hipSuccess) return bail(r, "hipEventCreate");
  if ((r = hipHostMalloc((void**)&ctx->h_row_cost, sizeof(uint32_t) * ROW_COST_MAX * SvrContext::MAX_OPS, hipHostMallocDefault)) != hipSuccess)
    return bail(r, "hipHostMalloc");
  std::memset(ctx->h_row_cost, 0, sizeof(uint32_t) * ROW_COST_MAX * SvrContext::MAX_OPS);
  if ((r = hipHostMalloc((void**)&ctx->h_failed_seq, 64, hipHostMallocDefault)) != hipSuccess) return bail(r, "hipHostMalloc");
  *ctx->h_failed_seq = 0;
  if ((r = hipMalloc((void**)&ctx->d_poison, 256)) != hipSuccess) return bail(r, "hipMalloc");
  if ((r = hipMemset(ctx->d_poison, 0, 256)) != hipSuccess) return bail(r, "hipMemset");
  *out = ctx;
  return SVR_OK;
}

void svr_destroy(SvrContext* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->gstream) (void)hipStreamSynchronize(ctx->gstream);
  if (ctx->gstream_hi) (void)hipStreamSynchronize(ctx->gstream_hi);
  for (auto& m : ctx->meshes) {
    if (m.vtx) (void)hipFree(m.vtx);
    if (m.idx) (void)hipFree(m.idx);
    if (m.groups) (void)hipFree(m.groups);
  }
  if (ctx->tex_arena) (void)hipFree(ctx->tex_arena);
  DevBuf* bufs[] = {&ctx->tex_table, &ctx->d_cvt, &ctx->d_trace, &ctx->d_tile_cycles, &ctx->mesh_table, &ctx->mat_table};
  for (auto& set : ctx->sets) {
    DevBuf* sb[] = {&set.inputs, &set.recs, &set.clipq, &set.bigq, &set.tiles, &set.bins, &set.pairs, &set.flat, &set.sorta};
    for (DevBuf* b : sb) b->release();
    if (set.ev_bin) (void)hipEventDestroy(set.ev_bin);
  }
  if (ctx->gstream) (void)hipStreamDestroy(ctx->gstream);
  if (ctx->gstream_hi) (void)hipStreamDestroy(ctx->gstream_hi);
  if (ctx->ev_gswitch) (void)hipEventDestroy(ctx->ev_gswitch);
  for (DevBuf* b : bufs) b->release();
  for (int i = 0; i < SvrContext::MAX_OPS; i++)
    if (ctx->h_stage[i]) (void)hipHostFree(ctx->h_stage[i]);
  if (ctx->h_counters) (void)hipHostFree(ctx->h_counters);
  if (ctx->h_row_cost) (void)hipHostFree(ctx->h_row_cost);
  for (int i = 0; i < SvrContext::MAX_OPS; i++)
    if (ctx->op_done[i]) (void)hipEventDestroy(ctx->op_done[i]);
  for (int i = 0; i < SvrContext::MAX_OPS; i++)
    if (ctx->op_start[i]) (void)hipEventDestroy(ctx->op_start[i]);
  if (ctx->d_poison) (void)hipFree(ctx->d_poison);
  if (ctx->h_failed_seq) (void)hipHostFree(ctx->h_failed_seq);
  for (int i = 0; i < SvrContext::TRING; i++)
    for (int k = 0; k < 5; k++)
      if (ctx->tev[i][k]) (void)hipEventDestroy(ctx->tev[i][k]);
  if (ctx->color_own) (void)hipFree(ctx->color_own);
  if (ctx->depth_own) (void)hipFree(ctx->depth_own);
  delete ctx;
}

int svr_set_stream(SvrContext* ctx, void* hip_stream) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;  // work on the old stream must not be orphaned
  ctx->stream = (hipStream_t)hip_stream;
  return SVR_OK;
}

int svr_bind_targets(SvrContext* ctx, void* color_dev, void* depth_dev) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if ((color_dev == nullptr) != (depth_dev == nullptr))
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_bind_targets: pass both targets or both NULL");
  if (int e = use_device(ctx)) return e;
  // no fence: passes already enqueued carry their own target pointers (also for a replay), so frames
  // can alternate between target sets while earlier ones are still in flight
  if (int e = poll_pending(ctx)) return e;
  if (int e = flush_clear(ctx)) return e;  // a deferred clear belongs to the targets it was asked for
  ctx->color = color_dev ? color_dev : ctx->color_own;
  ctx->depth = depth_dev ? (float*)depth_dev : ctx->depth_own;
  return SVR_OK;
}

int svr_get_targets(SvrContext* ctx, void** color_dev, void** depth_dev) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (int e = use_device(ctx)) return e;
  if (int e = flush_clear(ctx)) return e;  // the caller is about to look at the memory itself
  if (color_dev) *color_dev = ctx->color;
  if (depth_dev) *depth_dev = ctx->depth;
  return SVR_OK;
}

int svr_upload_mesh(SvrContext* ctx, const uint32_t* indices, size_t n_indices, const SvrVertex* vertices,
                    size_t n_vertices, SvrMesh* out) {
  if (!ctx || !out || (!indices && n_indices) || (!vertices && n_vertices))
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_upload_mesh: null argument");
  if (n_vertices > 0xffffffffull || n_indices > 0xffffffffull)
    return fail(SVR_ERR_UNSUPPORTED, "svr_upload_mesh: more than 2^32 elements");
  for (size_t i = 0; i < n_indices; i++)
    if (indices[i] >= n_vertices) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_upload_mesh: index out of range");
  if (int e = use_device(ctx)) return e;
  MeshRes m;
  // device-local vertex + index buffers, blocking staged copy (src/vk_engine.cpp:345-381)
  HIPCHK(hipMalloc((void**)&m.vtx, std::max<size_t>(n_vertices * sizeof(SvrVertex), 64)));
  hipError_t r = hipMalloc((void**)&m.idx, std::max<size_t>(n_indices * 4, 64));
  if (r != hipSuccess) {
    (void)hipFree(m.vtx);
    return fail(SVR_ERR_OUT_OF_MEMORY, std::string("hipMalloc(indices): ") + hipGetErrorString(r));
  }
  if (n_vertices) HIPCHK(hipMemcpy(m.vtx, vertices, n_vertices * sizeof(SvrVertex), hipMemcpyHostToDevice));
  if (n_indices) HIPCHK(hipMemcpy(m.idx, indices, n_indices * 4, hipMemcpyHostToDevice));
  // Index-group boxes: min / max position of the vertices named by every 192 consecutive indices.  A wave of the
  // setup kernel handles 64 consecutive triangles of a draw, i.e. at most two such groups, and skips them when
  // their box cannot reach the scissor.  (The bounds a caller attaches to a RenderObject are the loader's —
  // src/vk_loader.cpp:366-375, over all vertices of the mesh so far — and only is_visible may trust them.)
  {
    const size_t n_groups = (n_indices + GROUP_INDICES - 1) / GROUP_INDICES;
    std::vector<float> boxes(std::max<size_t>(n_groups, 1) * GROUP_WORDS);
    for (size_t g = 0; g < n_groups; g++) {
      float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
      uint32_t vmin = 0xffffffffu, vmax = 0u;  // the vertices the group names: the setup kernel stages that range through LDS
      const size_t end = std::min(n_indices, (g + 1) * GROUP_INDICES);
      for (size_t i = g * GROUP_INDICES; i < end; i++) {
        vmin = std::min(vmin, indices[i]);
        vmax = std::max(vmax, indices[i]);
        const float* p = vertices[indices[i]].position;
        for (int k = 0; k < 3; k++) {
          lo[k] = p[k] < lo[k] ? p[k] : lo[k];
          hi[k] = p[k] > hi[k] ? p[k] : hi[k];
          if (!std::isfinite(p[k])) lo[k] = hi[k] = NAN;  // poisons the box for good (comparisons with NaN are false): the kernel never culls on it
        }
      }
      for (int k = 0; k < 3; k++) {
        boxes[g * GROUP_WORDS + k] = lo[k];
        boxes[g * GROUP_WORDS + 3 + k] = hi[k];
      }
      std::memcpy(&boxes[g * GROUP_WORDS + 6], &vmin, 4);
      std::memcpy(&boxes[g * GROUP_WORDS + 7], &vmax, 4);
    }
    hipError_t rg = hipMalloc((void**)&m.groups, boxes.size() * sizeof(float));
    if (rg == hipSuccess) rg = hipMemcpy(m.groups, boxes.data(), boxes.size() * sizeof(float), hipMemcpyHostToDevice);
    if (rg != hipSuccess) {
      (void)hipFree(m.vtx);
      (void)hipFree(m.idx);
      if (m.groups) (void)hipFree(m.groups);
      return fail(SVR_ERR_OUT_OF_MEMORY, std::string("svr_upload_mesh: index-group boxes: ") + hipGetErrorString(rg));
    }
  }
  m.n_vtx = n_vertices;
  m.n_idx = n_indices;
  m.alive = true;
  ctx->meshes.push_back(m);
  *out = (SvrMesh)ctx->meshes.size();
  return SVR_OK;
}

int svr_destroy_mesh(SvrContext* ctx, SvrMesh mesh) {
  MeshRes* m = ctx ? get_mesh(ctx, mesh) : nullptr;
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_destroy_mesh: bad handle");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  (void)hipFree(m->vtx);
  (void)hipFree(m->idx);
  (void)hipFree(m->groups);
  m->vtx = nullptr;
  m->idx = nullptr;
  m->groups = nullptr;
  m->alive = false;
  return SVR_OK;
}

int svr_create_image(SvrContext* ctx, const void* rgba8, uint32_t width, uint32_t height, int mipmapped,
                     SvrImage* out) {
  if (!ctx || !rgba8 || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_image: null argument");
  if (width == 0 || height == 0 || width > 16384 || height > 16384)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_image: extent must be in 1..16384");
  if (int e = use_device(ctx)) return e;
  ImageRes im;
  im.w = width;
  im.h = height;
  im.levels = 1;
  if (mipmapped) {  // src/vk_engine.cpp:1543-1545
    uint32_t m = std::max(width, height);
    while (m > 1) {
      m >>= 1;
      im.levels++;
    }
  }
  // Mip levels are laid out as if the image were padded to 2^lw x 2^lh, so a shader derives a
  // level's offset from (lw, lh, level) alone (svr::mip_offset) and needs no per-image table.
  while ((1u << im.lw) < width) im.lw++;
  while ((1u << im.lh) < height) im.lh++;
  for (uint32_t l = 0; l < im.levels; l++) im.off[l] = mip_offset(im.lw, im.lh, l);
  uint32_t lw = width, lh = height;
  size_t total = (size_t)mip_offset(im.lw, im.lh, im.levels - 1) +
                 (size_t)std::max(1u, width >> (im.levels - 1)) * std::max(1u, height >> (im.levels - 1)) * 4;
  if (im.lw + im.lh > 28) return fail(SVR_ERR_UNSUPPORTED, "svr_create_image: image larger than 1 GiB");
  im.bytes = std::max<size_t>(total, 256);
  if (int e = arena_alloc(ctx, im.bytes, &im.arena_off)) return e;
  uint8_t* base = ctx->tex_arena + im.arena_off;
  hipError_t r = hipMemcpy(base, rgba8, (size_t)width * height * 4, hipMemcpyHostToDevice);
  if (r != hipSuccess) {
    arena_free(ctx, im.arena_off, im.bytes);
    return fail(SVR_ERR_DEVICE, std::string("hipMemcpy(image): ") + hipGetErrorString(r));
  }
  // generate_mipmaps: level n -> n+1, each a 2:1 linear blit (src/vk_images.cpp:66-133)
  lw = width;
  lh = height;
  for (uint32_t l = 1; l < im.levels; l++) {
    uint32_t dw = std::max(1u, lw >> 1), dh = std::max(1u, lh >> 1);
    launch_downsample(base + im.off[l - 1], lw, lh, base + im.off[l], dw, dh, ctx->stream);
    lw = dw;
    lh = dh;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));  // immediate_submit is blocking (src/vk_engine.cpp:1110-1129)
  im.alive = true;
  ctx->images.push_back(im);
  *out = (SvrImage)ctx->images.size();
  return SVR_OK;
}

int svr_destroy_image(SvrContext* ctx, SvrImage image) {
  ImageRes* im = ctx ? get_image(ctx, image) : nullptr;
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_destroy_image: bad handle");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  arena_free(ctx, im->arena_off, im->bytes);
  im->alive = false;
  return SVR_OK;
}

int svr_read_image_level(SvrContext* ctx, SvrImage image, uint32_t level, void* dst, size_t bytes, uint32_t* w,
                         uint32_t* h) {
  ImageRes* im = ctx ? get_image(ctx, image) : nullptr;
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_read_image_level: bad handle");
  if (level >= im->levels) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_image_level: no such level");
  uint32_t lw = std::max(1u, im->w >> level), lh = std::max(1u, im->h >> level);
  if (w) *w = lw;
  if (h) *h = lh;
  if (dst) {
    size_t need = (size_t)lw * lh * 4;
    if (bytes < need) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_image_level: buffer too small");
    if (int e = use_device(ctx)) return e;
    HIPCHK(hipMemcpy(dst, ctx->tex_arena + im->arena_off + im->off[level], need, hipMemcpyDeviceToHost));
  }
  return SVR_OK;
}

int svr_create_sampler(SvrContext* ctx, const SvrSamplerDesc* desc, SvrSampler* out) {
  if (!ctx || !desc || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: null argument");
  if ((desc->mag_filter | 1) != 1 || (desc->min_filter | 1) != 1 || (desc->mipmap_mode | 1) != 1)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: bad filter enum");
  if (!(desc->min_lod <= desc->max_lod)) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: min_lod > max_lod");
  ctx->samplers.push_back(*desc);
  *out = (SvrSampler)ctx->samplers.size();
  return SVR_OK;
}

int svr_write_material(SvrContext* ctx, int pass, const float color_factors[4], const float metal_rough_factors[4],
                       SvrImage color_image, SvrSampler color_sampler, SvrMaterial* out) {
  if (!ctx || !color_factors || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_write_material: null argument");
  if (!get_image(ctx, color_image)) return fail(SVR_ERR_BAD_HANDLE, "svr_write_material: bad image");
  if (color_sampler == 0 || color_sampler > ctx->samplers.size())
    return fail(SVR_ERR_BAD_HANDLE, "svr_write_material: bad sampler");
  if (pass < 0 || pass > 2) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_write_material: bad pass");
  if (int e = use_device(ctx)) return e;
  MaterialRes m;
  m.pass = pass;
  for (int i = 0; i < 4; i++) {
    m.cf[i] = color_factors[i];
    m.mr[i] = metal_rough_factors ? metal_rough_factors[i] : 0.0f;
  }
  m.image = color_image - 1;
  m.sampler = color_sampler - 1;
  ctx->materials.push_back(m);
  ctx->tex_slots = 0;  // table is stale; rebuilt lazily at the next draw
  *out = (SvrMaterial)ctx->materials.size();
  return SVR_OK;
}

int svr_clear_color(SvrContext* ctx, const float rgba[4]) {
  if (!ctx || !rgba) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_clear_color: null argument");
  if (int e = use_device(ctx)) return e;
  if (int e = poll_pending(ctx)) return e;
  uint64_t packed;
  if (ctx->fmt == SVR_COLOR_RGBA16F) {
    packed = (uint64_t)host_f32_to_f16(rgba[0]) | ((uint64_t)host_f32_to_f16(rgba[1]) << 16) |
             ((uint64_t)host_f32_to_f16(rgba[2]) << 32) | ((uint64_t)host_f32_to_f16(rgba[3]) << 48);
  } else {
    packed = 0;
    for (int k = 0; k < 4; k++) {
      float c = rgba[k];
      if (!(c == c)) c = 0.0f;
      c = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
      packed |= (uint64_t)(uint32_t)std::nearbyintf(c * 255.0f) << (8 * k);
    }
  }
  // whole rows of the scissor (a rank of the multi-GPU path only owns its band); deferred (flush_clear)
  if (int e = flush_clear(ctx)) return e;  // an older deferred clear cannot be skipped in general
  ctx->pending_clear.valid = true;
  ctx->pending_clear.target = ctx->color;
  ctx->pending_clear.y0 = ctx->sy;
  ctx->pending_clear.rows = ctx->sh;
  ctx->pending_clear.fmt = ctx->fmt;
  ctx->pending_clear.packed = packed;
  if (ctx->tuning & TUNE_NO_LAZY_CLEAR) return flush_clear(ctx);
  return SVR_OK;
}

int svr_draw_background(SvrContext* ctx, int effect, const float data[16]) {
  if (!ctx || !data) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_background: null argument");
  if (effect != SVR_BACKGROUND_GRADIENT && effect != SVR_BACKGROUND_SKY)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_background: unknown effect");
  if (int e = use_device(ctx)) return e;
  if (int e = poll_pending(ctx)) return e;
  if (int e = flush_clear(ctx)) return e;
  int slot = 0;
  if (int e = log_slot(ctx, &slot)) return e;
  ctx->log.emplace_back();
  SvrContext::LoggedOp& op = ctx->log.back();
  op.slot = slot;
  op.fill_kind = 1;
  op.target = ctx->color;
  op.clear_fmt = ctx->fmt;
  op.tw = ctx->W;
  op.th = ctx->H;
  op.y_first = ctx->sy;
  op.n_rows = ctx->sh;
  op.bg_effect = effect;
  std::memcpy(op.bg_data, data, sizeof(op.bg_data));
  return submit_clear(ctx, op);
}

static int blit_checks(SvrContext* ctx, const void* dst, uint32_t dw, uint32_t dh, int fmt, const char* who) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, std::string(who) + ": null argument");
  if (dw == 0 || dh == 0 || dw > 16384 || dh > 16384) return fail(SVR_ERR_INVALID_ARGUMENT, std::string(who) + ": extent must be in 1..16384");
  if (fmt != SVR_SWAPCHAIN_B8G8R8A8 && fmt != SVR_SWAPCHAIN_R8G8B8A8) return fail(SVR_ERR_INVALID_ARGUMENT, std::string(who) + ": unknown format");
  return SVR_OK;
}

int svr_copy_to_swapchain(SvrContext* ctx, void* dst_dev, uint32_t dw, uint32_t dh, int fmt) {
  if (int e = blit_checks(ctx, dst_dev, dw, dh, fmt, "svr_copy_to_swapchain")) return e;
  if (int e = use_device(ctx)) return e;
  if (int e = poll_pending(ctx)) return e;
  if (int e = flush_clear(ctx)) return e;
  int slot = 0;
  if (int e = log_slot(ctx, &slot)) return e;
  ctx->log.emplace_back();
  SvrContext::LoggedOp& op = ctx->log.back();
  op.slot = slot;
  op.fill_kind = 2;
  op.target = ctx->color;
  op.clear_fmt = ctx->fmt;
  op.tw = ctx->W;
  op.th = ctx->H;
  op.blit_dst = dst_dev;
  op.blit_w = dw;
  op.blit_h = dh;
  op.blit_fmt = fmt;
  // identity extent: the rows of the scissor (a rank of the multi-GPU path presents its band); scaled: everything
  const bool identity = dw == ctx->W && dh == ctx->H;
  op.y_first = identity ? ctx->sy : 0u;
  op.n_rows = identity ? ctx->sh : dh;
  op.blit_row_end = op.y_first + op.n_rows;
  if (identity && ctx->rstride > 1u) {  // a rank of the interleaved form presents its own tile rows, in place
    op.blit_rstride = ctx->rstride;
    op.blit_roff = ctx->roff;
    op.n_rows = owned_tile_rows(ctx) * TILE;
  }
  op.blit_status = ctx->present_status;
  return submit_clear(ctx, op);
}

int svr_read_swapchain(SvrContext* ctx, uint32_t dw, uint32_t dh, int fmt, void* dst_host, size_t bytes) {
  if (int e = blit_checks(ctx, dst_host, dw, dh, fmt, "svr_read_swapchain")) return e;
  if (bytes < (size_t)dw * dh * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_swapchain: buffer too small");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;  // the read-back is a fence
  if (int e = ctx->d_cvt.ensure((size_t)dw * dh * 4)) return e;
  launch_blit(ctx->color, ctx->fmt, ctx->W, ctx->H, ctx->d_cvt.p, dw, dh, 0, dh, fmt, ctx->d_poison, 1u, 0u, dh, nullptr, 0u, ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(dst_host, ctx->d_cvt.p, (size_t)dw * dh * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_set_scissor(SvrContext* ctx, uint32_t x, uint32_t y, uint32_t w, uint32_t h) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (w == 0 || h == 0 || (uint64_t)x + w > ctx->W || (uint64_t)y + h > ctx->H)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_set_scissor: rectangle outside the target");
  ctx->sx = x;
  ctx->sy = y;
  ctx->sw = w;
  ctx->sh = h;
  return SVR_OK;
}

int svr_set_row_interleave(SvrContext* ctx, uint32_t stride, uint32_t offset) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (stride == 0 || stride > 64 || offset >= stride) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_set_row_interleave: need 1 <= stride <= 64, offset < stride");
  ctx->rstride = stride;
  ctx->roff = offset;
  return SVR_OK;
}

int svr_set_present_status(SvrContext* ctx, uint32_t* status_dev) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  ctx->present_status = status_dev;
  return SVR_OK;
}

static int validate_object(SvrContext* ctx, const SvrRenderObject& o, bool transparent_list) {
  const char* which = transparent_list ? "transparent" : "opaque";
  MeshRes* m = get_mesh(ctx, o.mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, std::string("svr_draw_geometry: bad mesh handle in ") + which);
  if (o.material == 0 || o.material > ctx->materials.size())
    return fail(SVR_ERR_BAD_HANDLE, std::string("svr_draw_geometry: bad material handle in ") + which);
  if ((uint64_t)o.first_index + o.index_count > m->n_idx)
    return fail(SVR_ERR_INVALID_ARGUMENT, std::string("svr_draw_geometry: index range outside the mesh in ") + which);
  // MeshNode::Draw routes by pass_type (src/vk_engine.cpp:1729-1733)
  bool is_tr = ctx->materials[o.material - 1].pass == SVR_PASS_TRANSPARENT;
  if (is_tr && !transparent_list)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: Transparent material in the opaque list");
  if (!is_tr && transparent_list)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: non-Transparent material in the transparent list");
  return SVR_OK;
}

int svr_draw_geometry(SvrContext* ctx, const SvrSceneData* scene, const SvrRenderObject* opaque, size_t n_opaque,
                      const SvrRenderObject* transparent, size_t n_transparent, SvrStats* out_stats) {
  if (!ctx || !scene || (!opaque && n_opaque) || (!transparent && n_transparent))
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: null argument");
  auto t0 = std::chrono::steady_clock::now();
  if (int e = use_device(ctx)) return e;
  for (size_t i = 0; i < n_opaque; i++)
    if (int e = validate_object(ctx, opaque[i], false)) return e;
  for (size_t i = 0; i < n_transparent; i++)
    if (int e = validate_object(ctx, transparent[i], true)) return e;
  if (owned_tile_rows(ctx) == 0) {  // interleaved rows and more ranks than tile rows: this one owns nothing
    ctx->pending_clear.valid = false;
    ctx->stats = SvrStats{};
    if (out_stats) *out_stats = ctx->stats;
    return SVR_OK;
  }
  if (ctx->tex_slots != ctx->materials.size() + 1)
    if (int e = upload_tex_table(ctx, nullptr)) return e;
  // Many objects: cull, sort and the per-object records run on the device (k_flatten.hip).  The three
  // counts of the stats then only exist after the pass (svr_get_stats); out_stats gets what the host knows.
  const size_t n_objects = n_opaque + n_transparent;
  const bool fits = n_objects <= 16384 && ctx->meshes.size() < (1u << 20) && ctx->materials.size() < (1u << 20);
  if (fits && n_objects > 0 && (ctx->device_flatten == 1 || (ctx->device_flatten == 0 && n_objects >= 2048))) {
    uint64_t tris_max = 0;
    size_t chunks_max = 0;
    for (size_t i = 0; i < n_objects; i++) {
      uint32_t t = (i < n_opaque ? opaque[i] : transparent[i - n_opaque]).index_count / 3u;
      tris_max += t;
      chunks_max += chunk_count((i < n_opaque ? opaque[i] : transparent[i - n_opaque]).first_index, t);
    }
    SvrStats st{};
    int e = run_pass_flatten(ctx, scene, opaque, n_opaque, transparent, n_transparent, tris_max, chunks_max);
    auto t1 = std::chrono::steady_clock::now();
    st.mesh_draw_time = std::chrono::duration<float, std::milli>(t1 - t0).count();
    ctx->stats = st;
    if (out_stats) *out_stats = st;
    return e;
  }
  // cull: opaque only (src/vk_engine.cpp:1361-1367)
  std::vector<uint32_t> order;
  order.reserve(n_opaque);
  for (size_t i = 0; i < n_opaque; i++)
    if (is_visible(opaque[i], scene->viewproj)) order.push_back((uint32_t)i);
  // sort (src/vk_engine.cpp:1369-1378): deterministic key (material, mesh, submission index)
  std::stable_sort(order.begin(), order.end(), [&](uint32_t ia, uint32_t ib) {
    const SvrRenderObject& a = opaque[ia];
    const SvrRenderObject& b = opaque[ib];
    if (a.material == b.material) return a.mesh < b.mesh;
    return a.material < b.material;
  });
  std::vector<DrawDesc> draws;
  draws.reserve(order.size() + n_transparent);
  SvrStats st{};
  auto push = [&](const SvrRenderObject& o) {
    const MeshRes& m = ctx->meshes[o.mesh - 1];
    const MaterialRes& mat = ctx->materials[o.material - 1];
    DrawDesc d;
    std::memset(&d, 0, sizeof(d));
    std::memcpy(d.mat, o.transform, 64);
    std::memcpy(d.color_factors, mat.cf, 16);
    d.vtx = m.vtx;
    d.idx = m.idx + o.first_index;
    d.groups = m.groups;
    d.first_index = o.first_index;
    d.tri_count = o.index_count / 3;
    d.tex = o.material - 1;
    d.flags = ((uint32_t)PIPE_MESH << F_KIND_SHIFT) | (mat.pass == SVR_PASS_TRANSPARENT ? F_TRANSPARENT : 0u);
    draws.push_back(d);
    st.drawcall_count++;
    st.triangle_count += (int)(o.index_count / 3);
  };
  for (uint32_t i : order) push(opaque[i]);
  for (size_t i = 0; i < n_transparent; i++) push(transparent[i]);
  st.culled_draws = (uint32_t)(n_opaque - order.size());
  int e = run_pass(ctx, scene, draws);
  auto t1 = std::chrono::steady_clock::now();
  st.mesh_draw_time = std::chrono::duration<float, std::milli>(t1 - t0).count();
  ctx->stats = st;
  if (out_stats) *out_stats = st;
  return e;
}

int svr_draw_colored_triangle(SvrContext* ctx, SvrStats* out_stats) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (owned_tile_rows(ctx) == 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_colored_triangle: this context owns no tile row (svr_set_row_interleave)");
  if (int e = use_device(ctx)) return e;
  if (ctx->tex_slots != ctx->materials.size() + 1)
    if (int e = upload_tex_table(ctx, nullptr)) return e;
  std::vector<DrawDesc> draws(1);
  std::memset(&draws[0], 0, sizeof(DrawDesc));
  draws[0].tri_count = 1;
  draws[0].flags = (uint32_t)PIPE_COLORED_TRIANGLE << F_KIND_SHIFT;
  SvrStats st{};
  st.drawcall_count = 1;
  st.triangle_count = 1;
  int e = run_pass(ctx, nullptr, draws);
  ctx->stats = st;
  if (out_stats) *out_stats = st;
  return e;
}

int svr_draw_tex_image(SvrContext* ctx, SvrMesh mesh, uint32_t first_index, uint32_t index_count,
                       const float render_matrix[16], SvrImage image, SvrSampler sampler, SvrStats* out_stats) {
  if (!ctx || !render_matrix) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_tex_image: null argument");
  MeshRes* m = get_mesh(ctx, mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad mesh");
  ImageRes* im = get_image(ctx, image);
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad image");
  if (sampler == 0 || sampler > ctx->samplers.size()) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad sampler");
  if ((uint64_t)first_index + index_count > m->n_idx)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_tex_image: index range outside the mesh");
  if (owned_tile_rows(ctx) == 0) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_tex_image: this context owns no tile row (svr_set_row_interleave)");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;  // the scratch binding slot is about to change
  TexBinding tb = make_binding(*im, ctx->samplers[sampler - 1]);
  if (int e = upload_tex_table(ctx, &tb)) return e;
  ctx->tex_slots = 0;  // scratch slot is in use: rebuild before the next mesh pass
  std::vector<DrawDesc> draws(1);
  DrawDesc& d = draws[0];
  std::memset(&d, 0, sizeof(d));
  std::memcpy(d.mat, render_matrix, 64);
  d.vtx = m->vtx;
  d.idx = m->idx + first_index;
  d.groups = m->groups;
  d.first_index = first_index;
  d.tri_count = index_count / 3;
  d.tex = (uint32_t)ctx->materials.size();
  d.flags = (uint32_t)PIPE_TEX_IMAGE << F_KIND_SHIFT;
  SvrStats st{};
  st.drawcall_count = 1;
  st.triangle_count = (int)(index_count / 3);
  int e = run_pass(ctx, nullptr, draws);
  ctx->stats = st;
  if (out_stats) *out_stats = st;
  return e;
}

int svr_run_mesh_vert(SvrContext* ctx, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices, const float world[16],
                      const SvrSceneData* scene, SvrMaterial material, float* out_clip, float* out_varyings) {
  if (!ctx || !world || !scene || !out_clip || !out_varyings)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_mesh_vert: null argument");
  MeshRes* m = get_mesh(ctx, mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_run_mesh_vert: bad mesh");
  if (material == 0 || material > ctx->materials.size()) return fail(SVR_ERR_BAD_HANDLE, "svr_run_mesh_vert: bad material");
  if ((uint64_t)first_vertex + n_vertices > m->n_vtx)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_mesh_vert: vertex range outside the mesh");
  if (n_vertices == 0) return SVR_OK;
  if (int e = use_device(ctx)) return e;
  float consts[36];
  std::memcpy(consts, world, 64);
  std::memcpy(consts + 16, scene->viewproj, 64);
  std::memcpy(consts + 32, ctx->materials[material - 1].cf, 16);
  float* d_consts = nullptr;
  float* d_out = nullptr;
  HIPCHK(hipMalloc((void**)&d_consts, sizeof(consts)));
  hipError_t r = hipMalloc((void**)&d_out, (size_t)n_vertices * 12 * sizeof(float));
  if (r != hipSuccess) {
    (void)hipFree(d_consts);
    return fail(SVR_ERR_OUT_OF_MEMORY, "svr_run_mesh_vert: hipMalloc failed");
  }
  int rc = SVR_OK;
  do {
    if (hipMemcpy(d_consts, consts, sizeof(consts), hipMemcpyHostToDevice) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
    float* d_clip = d_out;
    float* d_var = d_out + (size_t)n_vertices * 4;
    launch_mesh_vert(m->vtx, first_vertex, n_vertices, d_consts, d_consts + 16, d_consts + 32, d_clip, d_var, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "mesh_vert kernel failed"); break; }
    if (hipMemcpy(out_clip, d_clip, (size_t)n_vertices * 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
    if (hipMemcpy(out_varyings, d_var, (size_t)n_vertices * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
  } while (0);
  (void)hipFree(d_consts);
  (void)hipFree(d_out);
  return rc;
}

int svr_run_vertex_shader(SvrContext* ctx, int shader, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices,
                          const float render_matrix[16], float* out_clip, float* out_varyings) {
  if (!ctx || !out_clip || !out_varyings) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: null argument");
  MeshRes* m = nullptr;
  if (shader == SVR_VS_COLORED_TRIANGLE) {
    if ((uint64_t)first_vertex + n_vertices > 3) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: colored_triangle.vert has 3 vertices");
  } else if (shader == SVR_VS_COLORED_TRIANGLE_MESH) {
    if (!render_matrix) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: null matrix");
    m = get_mesh(ctx, mesh);
    if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_run_vertex_shader: bad mesh");
    if ((uint64_t)first_vertex + n_vertices > m->n_vtx)
      return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: vertex range outside the mesh");
  } else {
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: unknown shader");
  }
  if (n_vertices == 0) return SVR_OK;
  if (int e = use_device(ctx)) return e;
  float* d_buf = nullptr;  // 16 floats of matrix, then clip, then varyings
  if (hipMalloc((void**)&d_buf, (16 + (size_t)n_vertices * 12) * sizeof(float)) != hipSuccess)
    return fail(SVR_ERR_OUT_OF_MEMORY, "svr_run_vertex_shader: hipMalloc failed");
  int rc = SVR_OK;
  do {
    float zero[16] = {0};
    if (hipMemcpy(d_buf, render_matrix ? render_matrix : zero, 64, hipMemcpyHostToDevice) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
    float* d_clip = d_buf + 16;
    float* d_var = d_clip + (size_t)n_vertices * 4;
    launch_vertex_shader(m ? m->vtx : nullptr, first_vertex, n_vertices, d_buf, d_clip, d_var, ctx->stream);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "vertex shader kernel failed"); break; }
    if (hipMemcpy(out_clip, d_clip, (size_t)n_vertices * 16, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
    if (hipMemcpy(out_varyings, d_var, (size_t)n_vertices * 32, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(SVR_ERR_DEVICE, "hipMemcpy"); break; }
  } while (0);
  (void)hipFree(d_buf);
  return rc;
}

int svr_set_option(SvrContext* ctx, int option, int64_t value) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (option == SVR_OPT_COUNT_FRAGMENTS) {
    ctx->instrument = value != 0;
    return SVR_OK;
  }
  if (option == SVR_OPT_TUNING) {
    ctx->tuning = (uint32_t)value;
    return SVR_OK;
  }
  if (option == SVR_OPT_DEVICE_FLATTEN) {
    if (value < 0 || value > 2) return fail(SVR_ERR_INVALID_ARGUMENT, "SVR_OPT_DEVICE_FLATTEN: 0 auto, 1 always, 2 never");
    ctx->device_flatten = (int)value;
    return SVR_OK;
  }
  if (option == SVR_OPT_QUEUE_CAPS) {
    if (value < 0 || value > (1 << 30)) return fail(SVR_ERR_INVALID_ARGUMENT, "SVR_OPT_QUEUE_CAPS: out of range");
    if (int e = use_device(ctx)) return e;
    if (int e = finish_pending(ctx)) return e;
    ctx->debug_caps = (uint32_t)value;
    ctx->clip_cap = ctx->extra_cap = ctx->bin_cap = 0;  // the next pass sizes its queues afresh
    return SVR_OK;
  }
  if (option == SVR_OPT_TILE_CYCLES) {
    ctx->tile_cycles = value != 0;
    return SVR_OK;
  }
  if (option == SVR_OPT_KERNEL_TIMING) {
    if (int e = use_device(ctx)) return e;
    for (int i = 0; i < SvrContext::TRING; i++)
      if (int e = harvest_timing(ctx, i)) return e;
    ctx->acc_ms[0] = ctx->acc_ms[1] = ctx->acc_ms[2] = 0.0;
    ctx->acc_n = 0;
    ctx->kernel_timing = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    return SVR_OK;
  }
  return fail(SVR_ERR_INVALID_ARGUMENT, "svr_set_option: unknown option");
}

int svr_debug_trace_pixel(SvrContext* ctx, int x, int y) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (int e = use_device(ctx)) return e;
  if (x >= 0) {
    if (int e = finish_pending(ctx)) return e;
    if (int e = ctx->d_trace.ensure(64 * sizeof(float))) return e;
    HIPCHK(hipMemset(ctx->d_trace.p, 0, 64 * sizeof(float)));
  }
  ctx->trace_x = x;
  ctx->trace_y = y;
  return SVR_OK;
}

int svr_debug_read_trace(SvrContext* ctx, float out[64]) {
  if (!ctx || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_trace: null argument");
  if (!ctx->d_trace.p) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_trace: tracing was never enabled");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(out, ctx->d_trace.p, 64 * sizeof(float), hipMemcpyDeviceToHost));
  return SVR_OK;
}

int svr_debug_read_bins(SvrContext* ctx, uint32_t* counts, size_t capacity, uint32_t* n_tiles) {
  if (!ctx || !n_tiles) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_bins: null argument");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  *n_tiles = ctx->last.n_tiles;
  if (!counts) return SVR_OK;
  if (capacity < 2 * (size_t)ctx->last.n_tiles || !ctx->last.tile_count)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_bins: buffer too small or no pass yet");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(counts, ctx->last.tile_count, 2 * (size_t)ctx->last.n_tiles * 4, hipMemcpyDeviceToHost));
  return SVR_OK;
}

int svr_debug_read_tile_cycles(SvrContext* ctx, uint32_t* cycles, size_t capacity) {
  if (!ctx || !cycles) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_tile_cycles: null argument");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  if (!ctx->last.tile_cycles || capacity < 4 * (size_t)ctx->last.n_tiles)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_tile_cycles: SVR_OPT_TILE_CYCLES was off for the last pass or buffer too small");
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(cycles, ctx->last.tile_cycles, 16 * (size_t)ctx->last.n_tiles, hipMemcpyDeviceToHost));
  return SVR_OK;
}

int svr_get_row_costs(SvrContext* ctx, uint32_t* costs, size_t capacity, uint32_t* n_tile_rows, uint32_t* first_row, uint32_t* n_rows) {
  if (!ctx || !n_tile_rows) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_row_costs: null argument");
  if (int e = use_device(ctx)) return e;
  if (int e = poll_pending(ctx)) return e;  // validates whatever has finished; never waits
  *n_tile_rows = (uint32_t)ctx->row_cost.size();
  if (first_row) *first_row = ctx->row_cost_y0;
  if (n_rows) *n_rows = ctx->row_cost_rows;
  if (costs) {
    if (capacity < ctx->row_cost.size()) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_row_costs: buffer too small");
    std::memcpy(costs, ctx->row_cost.data(), ctx->row_cost.size() * sizeof(uint32_t));
  }
  return SVR_OK;
}

int svr_debug_rcp_sweep(SvrContext* ctx, int variant, uint64_t first, uint64_t count, uint64_t* mismatches, uint64_t* refined,
                        uint32_t first_bad[16]) {
  if (!ctx || !mismatches) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_rcp_sweep: null argument");
  if (variant < 0 || variant > 2 || first + count > (1ull << 32)) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_rcp_sweep: bad range or variant");
  if (int e = use_device(ctx)) return e;
  unsigned long long* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 19 * 8));
  unsigned long long h[19] = {};
  hipError_t r = hipMemset(d, 0, 19 * 8);
  if (r == hipSuccess) {
    launch_rcp_sweep(variant, first, count, d, ctx->stream);
    r = hipStreamSynchronize(ctx->stream);
  }
  if (r == hipSuccess) r = hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (r != hipSuccess) return fail(SVR_ERR_DEVICE, std::string("svr_debug_rcp_sweep: ") + hipGetErrorString(r));
  *mismatches = h[0];
  if (refined) *refined = h[1];
  if (first_bad)
    for (int k = 0; k < 16; k++) first_bad[k] = (uint32_t)h[2 + k];
  return SVR_OK;
}

int svr_sync(SvrContext* ctx) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (int e = use_device(ctx)) return e;
  if (int e = finish_pending(ctx)) return e;
  HIPCHK(hipStreamSynchronize(ctx->stream));
  return SVR_OK;
}

int svr_read_color(SvrContext* ctx, void* dst, size_t bytes, int as_rgba8) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: null argument");
  if (int e = svr_sync(ctx)) return e;
  size_t n = (size_t)ctx->W * ctx->H;
  if (ctx->fmt == SVR_COLOR_RGBA8 || !as_rgba8) {
    size_t need = n * (ctx->fmt == SVR_COLOR_RGBA8 ? 4 : 8);
    if (bytes < need) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: buffer too small");
    HIPCHK(hipMemcpy(dst, ctx->color, need, hipMemcpyDeviceToHost));
    return SVR_OK;
  }
  if (bytes < n * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: buffer too small");
  if (int e = ctx->d_cvt.ensure(n * 4)) return e;
  launch_rgba16f_to_rgba8(ctx->color, ctx->d_cvt.p, (uint32_t)n, ctx->stream);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(ctx->stream));
  HIPCHK(hipMemcpy(dst, ctx->d_cvt.p, n * 4, hipMemcpyDeviceToHost));
  return SVR_OK;
}

int svr_read_depth(SvrContext* ctx, float* dst, size_t bytes) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_depth: null argument");
  if (int e = svr_sync(ctx)) return e;
  size_t n = (size_t)ctx->W * ctx->H;
  if (bytes < n * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_depth: buffer too small");
  HIPCHK(hipMemcpy(dst, ctx->depth, n * 4, hipMemcpyDeviceToHost));
  return SVR_OK;
}

int svr_get_stats(SvrContext* ctx, SvrStats* out) {
  if (!ctx || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_stats: null argument");
  if (int e = svr_sync(ctx)) return e;
  for (int i = 0; i < SvrContext::TRING; i++)
    if (int e = harvest_timing(ctx, i)) return e;
  *out = ctx->stats;
  out->replayed_passes = ctx->replayed;
  out->timed_passes = ctx->acc_n;
  if (ctx->acc_n) {
    out->geometry_ms = (float)(ctx->acc_ms[0] / ctx->acc_n);
    out->binning_ms = (float)(ctx->acc_ms[1] / ctx->acc_n);
    out->tile_ms = (float)(ctx->acc_ms[2] / ctx->acc_n);
    out->gpu_time_ms = out->geometry_ms + out->binning_ms + out->tile_ms;
  }
  return SVR_OK;
}

}  // extern "C"
