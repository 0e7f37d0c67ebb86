// k_flatten.hip — the host half of draw_geometry as a device pass: is_visible
// culling (src/vk_engine.cpp:56-86, 1361-1367), the opaque sort (:1369-1378, deterministic key
// (material, mesh, submission index)) and the per-object draw records of the record lambda
// (:1412-1457), for the object counts where the host loop becomes the limit (configs[4]: 5408 objects
// = 0.26 ms of host time per frame, more than an eighth of that frame costs on eight GPUs).
//
//   cull_kernel     lane per opaque object: is_visible in the scalar GLM operation order (no fma:
//                   the library is built with -ffp-contract=off), 64-bit sort key or "culled"
//   rank_kernel     16 lanes per opaque object count the smaller keys (N^2 / 16 compares per lane
//                   group, N <= 16384: 2 us of the chip at 5408 objects); the rank IS the draw slot,
//                   so the same lanes write the DrawDesc; transparent objects follow in submission order
//   prefix_kernel   one workgroup: exclusive scans of triangle and wave-chunk counts over the draws
//   chunks_kernel   lane per draw: its WaveChunk records
// Outputs are the same DrawDesc[] / WaveChunk[] the host path stages, plus the actual counts in the
// pass's Counters (the host only knows upper bounds: buffers and grids are sized by those).
#include "svr_launch.h"

namespace svr {

__device__ __forceinline__ void glm_matmul(const float* a, const float* b, float* out) {
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 4; r++) {
      float acc = a[0 + r] * b[4 * j + 0];
      acc = acc + a[4 + r] * b[4 * j + 1];
      acc = acc + a[8 + r] * b[4 * j + 2];
      acc = acc + a[12 + r] * b[4 * j + 3];
      out[4 * j + r] = acc;
    }
}

// is_visible, operation for operation as svr_api.hip's host version (and the oracle's)
__device__ bool is_visible_dev(const SvrRenderObject& obj, const float* viewproj) {
  float m[16];
  glm_matmul(viewproj, obj.transform, m);
  float mn[3] = {1.5f, 1.5f, 1.5f}, mx[3] = {-1.5f, -1.5f, -1.5f};
  for (int c = 0; c < 8; c++) {
    float sx = (c & 4) ? -1.0f : 1.0f, sy = (c & 2) ? -1.0f : 1.0f, sz = (c & 1) ? -1.0f : 1.0f;
    float p0 = obj.bounds.origin[0] + sx * obj.bounds.extents[0];
    float p1 = obj.bounds.origin[1] + sy * obj.bounds.extents[1];
    float p2 = obj.bounds.origin[2] + sz * obj.bounds.extents[2];
    float v[4];
    for (int r = 0; r < 4; r++) {
      float add0 = m[0 + r] * p0 + m[4 + r] * p1;
      float add1 = m[8 + r] * p2 + m[12 + r] * 1.0f;
      v[r] = add0 + add1;
    }
    v[0] = v[0] / v[3];
    v[1] = v[1] / v[3];
    v[2] = v[2] / v[3];
    for (int k = 0; k < 3; k++) {
      mn[k] = (mn[k] < v[k]) ? mn[k] : v[k];
      mx[k] = (v[k] < mx[k]) ? mx[k] : v[k];
    }
  }
  return !(mn[2] > 1.f || mx[2] < 0.f || mn[0] > 1.f || mx[0] < -1.f || mn[1] > 1.f || mx[1] < -1.f);
}

constexpr unsigned long long KEY_CULLED = ~0ull;

// The objects lie in pinned host memory (108 bytes each).  Read object by object — a lane its own 108 bytes, here and
// again by rank_kernel's record writers — they crossed the host link twice in requests of a few bytes: 5408 objects,
// 100 us for the two kernels.  Every block first pulls ITS objects (opaque: 256 per block; the blocks behind them take
// the transparent list) into device memory as a run of dwords, lanes on consecutive words, and everything behind reads
// the device copy.
__global__ __launch_bounds__(256) void cull_kernel(FlattenParams F) {
  const uint32_t n_all = F.n_opaque + F.n_transparent;
  const uint32_t first = blockIdx.x * 256u, count = min(256u, n_all - min(first, n_all));
  {
    // (256 objects are 27 648 bytes: every block's run starts on a 16-byte boundary of the two 64-byte-aligned arrays,
    // so the bulk moves as 16-byte pieces — a kilobyte per wave instruction on the host link — and only the last
    // block's tail as words)
    static_assert((256u * sizeof(SvrRenderObject)) % 16u == 0u && sizeof(SvrRenderObject) % 4u == 0u, "aligned runs");
    const uint32_t bytes = count * (uint32_t)sizeof(SvrRenderObject), n16 = bytes / 16u;
    const uint4* src = reinterpret_cast<const uint4*>(F.objects + first);
    uint4* dst = reinterpret_cast<uint4*>(F.objects_dev + first);
    for (uint32_t w = threadIdx.x; w < n16; w += 256u) dst[w] = src[w];
    for (uint32_t w = n16 * 4u + threadIdx.x; w < bytes / 4u; w += 256u)
      reinterpret_cast<uint32_t*>(dst)[w] = reinterpret_cast<const uint32_t*>(src)[w];
  }
  __threadfence_block();
  __syncthreads();  // (a block reads back only what it wrote itself: the same CU's stores)
  uint32_t i = first + threadIdx.x;
  bool live = i < F.n_opaque, vis = false;
  if (live) {
    const SvrRenderObject& o = F.objects_dev[i];
    vis = is_visible_dev(o, F.viewproj);
    F.keys[i] = vis ? (((unsigned long long)o.material << 44) | ((unsigned long long)o.mesh << 24) | (unsigned long long)i) : KEY_CULLED;
  }
  unsigned long long mv = __ballot(vis), mc = __ballot(live && !vis);
  if ((threadIdx.x & 63u) == 0) {
    if (mv) atomicAdd(&F.counters->flat_draws, (uint32_t)__popcll(mv));  // visible opaque so far; prefix_kernel adds the transparent ones
    if (mc) atomicAdd(&F.counters->flat_culled, (uint32_t)__popcll(mc));
  }
}

__device__ __forceinline__ void write_draw(const FlattenParams& F, uint32_t slot, const SvrRenderObject& o) {
  const MeshEntry me = F.meshes[o.mesh - 1];
  const MatEntry ma = F.materials[o.material - 1];
  DrawDesc d;
#pragma unroll
  for (int k = 0; k < 16; k++) d.mat[k] = o.transform[k];
#pragma unroll
  for (int k = 0; k < 4; k++) d.color_factors[k] = ma.cf[k];
  matmul4(F.viewproj, d.mat, d.mvp);  // C0, once per draw
  d.vtx = me.vtx;
  d.idx = me.idx + o.first_index;
  d.groups = me.groups;
  d.first_index = o.first_index;
  d.pad = 0;
  d.tri_count = o.index_count / 3u;
  d.tri_base = 0;  // prefix_kernel
  d.tex = o.material - 1u;
  d.flags = ((uint32_t)PIPE_MESH << F_KIND_SHIFT) | (ma.pass == SVR_PASS_TRANSPARENT ? F_TRANSPARENT : 0u);
  F.draws[slot] = d;
  F.draw_tris[slot] = d.tri_count;
  F.chunk_base[slot] = chunk_count(d.first_index, d.tri_count);  // the count for now: prefix_kernel turns it into the base
}

// 16 lanes per opaque object (lane & 15 = the part of the key array it scans); transparent objects: one lane each.
// The keys go through LDS 2048 at a time (a lane's 338 global loads, one per round of its loop, were the kernel: 86 us
// at 5408 objects): the block's 16 objects x 16 parts read consecutive keys, the four groups of a wave the same ones.
__global__ __launch_bounds__(256) void rank_kernel(FlattenParams F) {
  constexpr uint32_t KT = 2048;
  __shared__ unsigned long long s_keys[KT];
  const uint32_t group = (blockIdx.x * blockDim.x + threadIdx.x) >> 4, part = threadIdx.x & 15u;
  const uint32_t n_vis = F.counters->flat_draws;
  if (blockIdx.x * 16u < F.n_opaque) {  // a block of opaque groups (block-uniform; its last groups may lie beyond the list)
    const bool mine_ok = group < F.n_opaque;
    const unsigned long long mine = mine_ok ? F.keys[group] : KEY_CULLED;
    uint32_t smaller = 0;
    for (uint32_t base = 0; base < F.n_opaque; base += KT) {
      const uint32_t nk = min(KT, F.n_opaque - base);
      __syncthreads();
      for (uint32_t j = threadIdx.x; j < nk; j += 256u) s_keys[j] = F.keys[base + j];
      __syncthreads();
      if (mine != KEY_CULLED)
        for (uint32_t j = part; j < nk; j += 16u) smaller += s_keys[j] < mine ? 1u : 0u;
    }
    smaller += __shfl_xor(smaller, 1);
    smaller += __shfl_xor(smaller, 2);
    smaller += __shfl_xor(smaller, 4);
    smaller += __shfl_xor(smaller, 8);
    if (part == 0 && mine != KEY_CULLED) write_draw(F, smaller, F.objects_dev[group]);
  } else {
    // the blocks behind the opaque ones: 256 transparent objects each, a lane per object
    const uint32_t opaque_blocks = (F.n_opaque + 15u) / 16u;
    const uint32_t t = (blockIdx.x - opaque_blocks) * 256u + threadIdx.x;
    if (t < F.n_transparent) write_draw(F, n_vis + t, F.objects_dev[F.n_opaque + t]);
  }
}

// one workgroup: tri_base and chunk_base of every draw (<= 16 per thread), totals into the counters
__global__ __launch_bounds__(1024) void prefix_kernel(FlattenParams F) {
  __shared__ uint32_t s_tri[16], s_chk[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t n = F.counters->flat_draws + F.n_transparent;
  const uint32_t per = (n + 1023u) / 1024u, first = tid * per;
  uint32_t tri = 0, chk = 0;
  for (uint32_t k = 0; k < per; k++)
    if (first + k < n) {
      tri += F.draw_tris[first + k];
      chk += F.chunk_base[first + k];  // (rank_kernel left the draw's chunk count here)
    }
  uint32_t itri = tri, ichk = chk;
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t a = __shfl_up(itri, off), b = __shfl_up(ichk, off);
    if ((int)lane >= off) {
      itri += a;
      ichk += b;
    }
  }
  if (lane == 63) {
    s_tri[wv] = itri;
    s_chk[wv] = ichk;
  }
  __syncthreads();
  uint32_t btri = 0, bchk = 0, ttri = 0, tchk = 0;
  for (uint32_t w = 0; w < 16; w++) {
    if (w < wv) {
      btri += s_tri[w];
      bchk += s_chk[w];
    }
    ttri += s_tri[w];
    tchk += s_chk[w];
  }
  uint32_t rt = btri + itri - tri, rc = bchk + ichk - chk;
  for (uint32_t k = 0; k < per; k++)
    if (first + k < n) {
      const uint32_t t = F.draw_tris[first + k], c = F.chunk_base[first + k];
      F.draws[first + k].tri_base = rt;
      F.chunk_base[first + k] = rc;
      rt += t;
      rc += c;
    }
  if (tid == 0) {
    F.counters->flat_draws = n;
    F.counters->flat_tris = ttri;
    F.counters->flat_chunks = tchk;
  }
}

__global__ __launch_bounds__(256) void chunks_kernel(FlattenParams F) {
  uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= F.counters->flat_draws) return;
  uint32_t t = F.draw_tris[r], base = F.chunk_base[r];
  const uint32_t fi = F.draws[r].first_index;
  for (uint32_t c = 0, nc = chunk_count(fi, t); c < nc; c++) {
    WaveChunk ch;
    ch.draw = r;
    ch.first_tri = chunk_first(fi, c);
    F.chunks[base + c] = ch;
  }
}

void launch_flatten(const FlattenParams& F, hipStream_t s) {
  const uint32_t n_all = F.n_opaque + F.n_transparent;
  if (n_all == 0) return;
  hipLaunchKernelGGL(cull_kernel, dim3((n_all + 255u) / 256u), dim3(256), 0, s, F);  // (also pulls every object into device memory)
  hipLaunchKernelGGL(rank_kernel, dim3((F.n_opaque + 15u) / 16u + (F.n_transparent + 255u) / 256u), dim3(256), 0, s, F);
  hipLaunchKernelGGL(prefix_kernel, dim3(1), dim3(1024), 0, s, F);
  hipLaunchKernelGGL(chunks_kernel, dim3((n_all + 255u) / 256u), dim3(256), 0, s, F);
}

}  // namespace svr
