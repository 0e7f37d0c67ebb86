// k_geometry.hip — vertex stage + primitive assembly + clip + triangle setup.
//
// Replaces, per vkCmdDrawIndexed (src/vk_engine.cpp:1453): index fetch, mesh.vert
// (shaders/mesh.vert:29-38), clipping to the Vulkan clip volume, perspective divide, viewport
// transform (src/vk_engine.cpp:1421-1429) and the rasteriser's triangle setup.
//
// One lane per triangle, one wave per 64 consecutive triangles of ONE draw (WaveChunk), so the
// draw's matrices are wave-uniform.  A chunk's vertices are read as they lie — the index-group table names
// the run of the vertex buffer its 192 indices touch, consecutive lanes read consecutive 48-byte vertices —
// run through mesh.vert ONCE each and parked in LDS for the chunk's triangles to pick up (runs longer than 128
// vertices fall back to a gather per corner); no post-transform buffer goes through memory.  Triangles that
// need real clipping (any vertex beyond the near/far planes or the guard band) are queued and handled by the
// clipper blocks of the binning launch (k_bin.hip clip_and_bin) so this kernel keeps no polygon arrays in scratch.
#include <algorithm>

#include "svr_bin.h"
#include "svr_clip.h"
#include "svr_launch.h"

namespace svr {

// Can any fragment of the box [lo, hi] (object space) under clip = mvp * (p, 1) land inside the scissor?  Half-space
// tests in clip space, no division: every visible point lies inside the clip volume (w > 0 there), where
// "pixel row >= r" reads y >= (2 r / H - 1) w — linear, so a box whose eight corners all fail one of the tests
// fails it everywhere.  The scissor planes sit one pixel outside the scissor (snapping and rounding move a
// vertex by far less).  Lanes 0..7 take one corner each; the verdict is wave-uniform.  NaNs never cull.
__device__ __forceinline__ bool chunk_can_be_seen(const FrameParams& P, const float* mvp, const float lo[3], const float hi[3], uint32_t lane) {
  const float px = (lane & 4u) ? hi[0] : lo[0], py = (lane & 2u) ? hi[1] : lo[1], pz = (lane & 1u) ? hi[2] : lo[2];
  float c[4];
  matvec4(mvp, px, py, pz, 1.0f, c);
  const float inv_hw = 2.0f / (float)P.W, inv_hh = 2.0f / (float)P.H;
  const float x_lo = ((float)P.sx - 1.0f) * inv_hw - 1.0f, x_hi = ((float)(P.sx + P.sw) + 1.0f) * inv_hw - 1.0f;
  const float y_lo = ((float)P.sy - 1.0f) * inv_hh - 1.0f, y_hi = ((float)(P.sy + P.sh) + 1.0f) * inv_hh - 1.0f;
  const unsigned long long corners = 0xffull;
  const bool out_near = c[2] > c[3], out_far = c[2] < 0.0f;
  const bool out_l = c[0] < x_lo * c[3], out_r = c[0] > x_hi * c[3];
  const bool out_t = c[1] < y_lo * c[3], out_b = c[1] > y_hi * c[3];
  const bool gone = (__ballot(out_near) & corners) == corners || (__ballot(out_far) & corners) == corners ||
                    (__ballot(out_l) & corners) == corners || (__ballot(out_r) & corners) == corners ||
                    (__ballot(out_t) & corners) == corners || (__ballot(out_b) & corners) == corners;
  if (gone) return false;
  if (P.rstride > 1u) {
    // Interleaved rows (a rank of the multi-GPU form owns every rstride-th tile row and runs the WHOLE scene's vertex
    // stage): a chunk whose box lies in front of the eye (every corner w > 0: its image is inside the hull of the
    // corners' images) and between two of the rank's tile rows goes no further.  One division per corner, the rows one
    // pixel wider than the corners' images (snapping and rounding move a vertex by far less).  configs[4] at N = 8: a
    // chunk of 64 triangles spans about three tile rows, so three chunks of five stop here.
    const bool front = (__ballot(c[3] > 0.0f) & corners) == corners;
    if (front) {
      float ys = (c[1] / c[3] + 1.0f) * (0.5f * (float)P.H);
      float lo_y = ys, hi_y = ys;
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {  // lanes 0..7 hold the corners: their partners under xor 1, 2, 4 are corners too
        lo_y = fminf(lo_y, __shfl_xor(lo_y, m));
        hi_y = fmaxf(hi_y, __shfl_xor(hi_y, m));
      }
      lo_y = __shfl(lo_y, 0);
      hi_y = __shfl(hi_y, 0);
      if (lo_y == lo_y && hi_y == hi_y && fabsf(lo_y) < 1.0e6f && fabsf(hi_y) < 1.0e6f) {  // (NaNs and wild values never cull)
        int miny = max((int)floorf(lo_y) - 1, (int)P.sy), maxy = min((int)ceilf(hi_y) + 1, (int)(P.sy + P.sh) - 1);
        if (miny > maxy) return false;
        int l0, l1;
        local_tile_rows(P, miny, maxy, l0, l1);
        if (l0 > l1) return false;
      }
    }
  }
  return true;
}

__global__ __launch_bounds__(256, 4) void setup_kernel(FrameParams P) {
  // the wave's chunk and its draw are wave-uniform: held in SGPRs, so chunk -> draw record is two scalar round
  // trips (read as per-lane values they were a chain of six vector loads in front of the first index fetch)
  const uint32_t gw = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
  uint32_t lane = threadIdx.x & 63;
  // device-flattened passes: the grid was sized by the host's upper bound, the real count is on the device
  const uint32_t n_chunks = P.flatten ? P.counters->flat_chunks : P.n_chunks;
  bool live = gw < n_chunks;  // wave-uniform; dead waves only attend the barriers
  // a dead wave touches neither the chunk list nor a draw it names: with every object culled the device
  // flatten writes no chunk at all and chunks[0] holds whatever an earlier pass left there (draws[0] is
  // always inside the inputs allocation; nothing read from it is used by a dead wave)
  WaveChunk ch{0u, 0u};
  if (live) ch = P.chunks[gw];
  ch.draw = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.draw);
  ch.first_tri = (uint32_t)__builtin_amdgcn_readfirstlane((int)ch.first_tri);
  // the whole 192-byte record at once, through the scalar cache (constant address space: written by an earlier kernel)
  typedef const __attribute__((address_space(4))) uint32_t* const_words;
  DrawDesc d;
  {
    const_words src = (const_words)(const void*)(P.draws + ch.draw);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&d);
#pragma unroll
    for (uint32_t i = 0; i < sizeof(DrawDesc) / 4u; i++) dst[i] = src[i];
  }
  uint32_t kind = (d.flags >> F_KIND_SHIFT) & 3u;
  // Whole chunks that cannot reach the scissor — outside the frustum, or (a rank of the multi-GPU path renders a
  // band of rows) above or below the band — are dropped before a single index is fetched: the box of the chunk's
  // vertices comes from the mesh's index-group table (svr_upload_mesh).
  uint32_t v_first = 0, v_count = 0;  // wave-uniform
  if (live && d.groups && kind != PIPE_COLORED_TRIANGLE) {
    const uint32_t first = d.first_index + 3u * ch.first_tri, last = first + 3u * chunk_len(d.first_index, d.tri_count, ch.first_tri) - 1u;
    typedef const __attribute__((address_space(4))) float* const_floats;
    const_floats g0 = (const_floats)(const void*)(d.groups + GROUP_WORDS * (first / GROUP_INDICES));
    const_floats g1 = (const_floats)(const void*)(d.groups + GROUP_WORDS * (last / GROUP_INDICES));
    float a[6], b[6], lo[3], hi[3];
#pragma unroll
    for (int k = 0; k < 6; k++) {  // both boxes in one round of scalar loads
      a[k] = g0[k];
      b[k] = g1[k];
    }
    // ... and, with them, the range of vertices the chunk's indices name
    v_first = min(f2u(g0[6]), f2u(g1[6]));
    v_count = max(f2u(g0[7]), f2u(g1[7])) - v_first + 1u;
    bool boxes_ok = true;  // a box with a non-finite vertex behind it is stored as NaNs: never culled
#pragma unroll
    for (int k = 0; k < 3; k++) {
      boxes_ok = boxes_ok & (a[k] <= a[3 + k]) & (b[k] <= b[3 + k]);
      lo[k] = fminf(a[k], b[k]);
      hi[k] = fmaxf(a[3 + k], b[3 + k]);
    }
    if (boxes_ok) live = chunk_can_be_seen(P, d.mvp, lo, hi, lane);
  }
  // (a chunk ends at the next line of the mesh's index-group grid: svr_device.h chunk_len)
  uint32_t tri = (live && lane < chunk_len(d.first_index, d.tri_count, ch.first_tri)) ? ch.first_tri + lane : 0xffffffffu;
  uint32_t seq = d.tri_base + tri;
  __shared__ uint32_t s_tot[8];
  // per wave 6 KiB: the chunk's shaded vertices (128 x 48 B), then — 4 KiB of it — one half (8 pieces) of 32 of its 64
  // records at a time, for the transposed store.  24 KiB per workgroup = 20 of gfx950's 1280-byte LDS granules: the
  // workgroup fits the hole ONE retiring tile workgroup leaves (25 granules; LDS is handed out in contiguous ranges).  With
  // 8 KiB per wave (the 64 records' half at once) it needed 26 — and once the tile kernel really held five workgroups
  // per CU, stage 1 of the next frame starved beside it: geometry 0.03 -> 0.13 ms under overlap, every frame longer.
  __shared__ uint4 s_tr[4][64 * 6];
  // The chunk's vertices through LDS.  Its 192 indices name a short run of the vertex buffer (a mesh's triangles
  // are laid out near their vertices), known from the group table before a single index has arrived: the wave reads
  // that run as it lies — consecutive lanes consecutive 48-byte vertices, 3 KiB per instruction, in flight TOGETHER
  // with the index fetch instead of behind it — runs mesh.vert once per vertex (two rounds of 64 at most, where the
  // three corners of 64 triangles were three), and parks the results in the wave's (still idle) transposition
  // buffer, where the triangles pick their corners up.  A chunk whose run is longer than STAGE_VERTS gathers as before.
  constexpr uint32_t STAGE_VERTS = 128;
  static_assert(STAGE_VERTS * sizeof(VOut) <= 64 * 6 * sizeof(uint4), "staged vertices fit the wave's buffer");
  const uint32_t wave_in_group = threadIdx.x >> 6;
  VOut* s_v = reinterpret_cast<VOut*>(s_tr[wave_in_group]);
  const bool staged = live && v_count != 0u && v_count <= STAGE_VERTS && kind != PIPE_COLORED_TRIANGLE;  // wave-uniform
  if (staged) {
#pragma unroll
    for (uint32_t r = 0; r < STAGE_VERTS / 64u; r++) {
      const uint32_t j = r * 64u + lane;
      if (j < v_count) {
        VOut o;
        shade_corner(d, kind, d.mvp, v_first + j, o);
        s_v[j] = o;
      }
    }
    __builtin_amdgcn_wave_barrier();  // the wave's own LDS traffic is in order; this keeps the compiler from moving it
  }
  uint4 piece[16];
  piece[0] = make_uint4(1u, 0u, 0u, 0u);  // the invalid record: minx = 1 > maxx = 0
  TriGeom g{};
  bool ok = false;  // a valid record was written and is to be binned
  if (tri < d.tri_count) {
    VOut v0, v1, v2;
    if (kind == PIPE_COLORED_TRIANGLE) {
      colored_triangle_vert(0, v0);
      colored_triangle_vert(1, v1);
      colored_triangle_vert(2, v2);
    } else {
      const float* mvp = d.mvp;  // sceneData.viewproj * PushConstants.renderMatrix, computed once per draw
      uint32_t i0 = d.idx[3 * tri + 0], i1 = d.idx[3 * tri + 1], i2 = d.idx[3 * tri + 2];
      if (staged) {
        v0 = s_v[i0 - v_first];
        v1 = s_v[i1 - v_first];
        v2 = s_v[i2 - v_first];
      } else {
        shade_corner(d, kind, mvp, i0, v0);
        shade_corner(d, kind, mvp, i1, v1);
        shade_corner(d, kind, mvp, i2, v2);
      }
    }
    int c0 = outcode(v0.clip), c1 = outcode(v1.clip), c2 = outcode(v2.clip);
    bool to_clip = false;
    if ((c0 & c1 & c2) == 0) {
      to_clip = true;
      if (((c0 | c1 | c2) & (OC_NEAR | OC_FAR)) == 0) {
        float hw = (float)P.W * 0.5f, hh = (float)P.H * 0.5f;
        ScreenV s0 = to_screen(v0.clip, hw, hh), s1 = to_screen(v1.clip, hw, hh), s2 = to_screen(v2.clip, hw, hh);
        if (s0.ok && s1.ok && s2.ok) {
          to_clip = false;
          ok = setup_triangle(P, &v0, &v1, &v2, s0, s1, s2, make_key(seq, d.flags, P.tex[d.tex], false), d.flags, P.tex[d.tex], piece, &g);
        }
      }
    }
    if (!ok) piece[0] = make_uint4(1u, 0u, 0u, 0u);
    if (ok && P.instrument) atomicAdd(&P.counters->binned, 1ull);
    if (to_clip) {  // slow path: hand over to the clipper (it links its pieces from this slot)
      uint32_t slot = atomicAdd(&P.counters->n_clip, 1u);
      if (slot < P.clip_cap) {
        ClipItem it;
        it.draw = ch.draw;
        it.tri = tri;
        P.clip_queue[slot] = it;
      } else {
        atomicOr(&P.counters->overflow, 1u);
      }
    }
  }
  // Store the wave's 64 consecutive records.  A lane storing its own record touches 64 different
  // lines per instruction (256-byte stride); transposed through LDS, half a record at a time, eight
  // lanes write one 128-byte line together.  Culled triangles only get their 16-byte header.
  {
    const unsigned long long act = __ballot(tri < d.tri_count), okm = __ballot(ok);
    uint4* wave_recs = reinterpret_cast<uint4*>(P.recs + (d.tri_base + ch.first_tri));
    uint4* sl = s_tr[threadIdx.x >> 6];
    // s_tr is also where the chunk's shaded vertices were parked (as VOut): every lane's reads of them lie in front of
    // this line, and nothing but this barrier keeps the compiler from moving the stores below across them
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int half = 0; half < 2; half++) {
#pragma unroll
      for (uint32_t r = 0; r < 2u; r++) {  // records 32 r .. 32 r + 31
        __builtin_amdgcn_wave_barrier();   // (the wave's LDS operations retire in order; this pins the compiler's order)
        if ((lane >> 5) == r) {
#pragma unroll
          for (int i = 0; i < 8; i++) sl[(lane & 31u) * 8u + ((uint32_t)i ^ (lane & 7u))] = piece[half * 8 + i];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const uint32_t tl = (uint32_t)k * 8u + (lane >> 3), i = lane & 7u, t = 32u * r + tl;
          uint4 v = sl[tl * 8u + (i ^ (tl & 7u))];
          bool header = half == 0 && i == 0u;
          if ((act >> t) & 1ull)
            if (header || ((okm >> t) & 1ull)) wave_recs[t * 16u + (uint32_t)half * 8u + i] = v;
        }
      }
    }
  }
  // Emit the triangle's (bin, record) pairs now, while bbox and edge functions are in registers.
  // Triangles over 16 tiles are queued instead; bin_rest (k_bin.hip) walks those wave-wide.
  TileRange tr = tile_range(P, g.minx, g.miny, g.maxx, g.maxy, ok);
  {  // queue push, one atomic per wave
    bool big = ok && tr.nt > SMALL_MAX_TILES;
    unsigned long long m = __ballot(big);
    if (m) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&P.counters->n_big, (uint32_t)__popcll(m));
      base = __shfl(base, 0);
      if (big) P.big_queue[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = seq;
    }
  }
  EdgeSet e;
  e.A0 = g.A[0]; e.A1 = g.A[1]; e.A2 = g.A[2]; e.B0 = g.B[0]; e.B1 = g.B[1]; e.B2 = g.B[2];
  e.C0 = g.C[0]; e.C1 = g.C[1]; e.C2 = g.C[2];
  emit_small_pairs(P, ok && tr.nt <= SMALL_MAX_TILES, tr, g.minx, g.miny, g.maxx, g.maxy,
                   (d.flags & F_TRANSPARENT) ? P.n_tiles : 0u, e, seq, s_tot);
}

// ------------------------------------------------------------------------------------------------
// mesh.vert as a stand-alone operator: one vertex per lane, three coalesced 16-byte loads per lane
// (a wave reads 3 KiB contiguous), writes gl_Position and the 8 varyings.
__global__ __launch_bounds__(256) void mesh_vert_kernel(const SvrVertex* vtx, uint32_t first, uint32_t n,
                                                        const float* world16, const float* viewproj16,
                                                        const float* cf4, float* out_clip, float* out_var) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float world[16], viewproj[16], mvp[16], cf[4];
  for (int k = 0; k < 16; k++) {
    world[k] = world16[k];
    viewproj[k] = viewproj16[k];
  }
  for (int k = 0; k < 4; k++) cf[k] = cf4[k];
  matmul4(viewproj, world, mvp);
  VertexRaw v = load_vertex(vtx, first + i);
  VOut o;
  mesh_vert(v, mvp, world, cf, o);
  reinterpret_cast<float4*>(out_clip)[i] = make_float4(o.clip[0], o.clip[1], o.clip[2], o.clip[3]);
  reinterpret_cast<float4*>(out_var)[2 * i] = make_float4(o.attr[0], o.attr[1], o.attr[2], o.attr[3]);
  reinterpret_cast<float4*>(out_var)[2 * i + 1] = make_float4(o.attr[4], o.attr[5], o.attr[6], o.attr[7]);
}

// colored_triangle.vert (vtx == nullptr: the vertex index is all it reads) and colored_triangle_mesh.vert as operators
__global__ __launch_bounds__(256) void vertex_shader_kernel(const SvrVertex* vtx, uint32_t first, uint32_t n,
                                                            const float* matrix16, float* out_clip, float* out_var) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  VOut o;
  if (vtx) {
    float m[16];
    for (int k = 0; k < 16; k++) m[k] = matrix16[k];
    VertexRaw v = load_vertex(vtx, first + i);
    colored_triangle_mesh_vert(v, m, o);
  } else {
    colored_triangle_vert((int)(first + i), o);
  }
  reinterpret_cast<float4*>(out_clip)[i] = make_float4(o.clip[0], o.clip[1], o.clip[2], o.clip[3]);
  reinterpret_cast<float4*>(out_var)[2 * i] = make_float4(o.attr[0], o.attr[1], o.attr[2], o.attr[3]);
  reinterpret_cast<float4*>(out_var)[2 * i + 1] = make_float4(o.attr[4], o.attr[5], o.attr[6], o.attr[7]);
}

void launch_setup(const FrameParams& P, hipStream_t s) {
  if (P.n_chunks == 0) return;
  uint32_t blocks = (P.n_chunks + 3) / 4;
  hipLaunchKernelGGL(setup_kernel, dim3(blocks), dim3(256), 0, s, P);
}
// Pass prologue: pull the pass inputs (DrawDesc[] + WaveChunk[], ~100 KB) out of the pinned staging
// buffer and zero the pass's counters, in one kernel.  A hipMemcpyAsync here is an SDMA packet with
// ~20 us of signalling between kernels on the critical chain of every pass, a hipMemsetAsync another
// launch; the staging memory is device-visible, so sixteen bytes per lane over the host link do it.
// The copy also fills in DrawDesc::mvp = viewproj * mat of the first n_draws records (MESH draws; others get mat):
// 16-byte piece j + 4 of a record is column j of the product, made from piece j by the thread that would copy it.
__global__ __launch_bounds__(256) void prologue_kernel(const uint4* host_src, uint4* dst, uint32_t n_copy, uint4* zero,
                                                       uint32_t n_zero, uint32_t n_draws, SvrSceneData scene) {
  const uint32_t stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
  constexpr uint32_t PIECES = sizeof(DrawDesc) / 16u;
  for (uint32_t i = t; i < n_copy; i += stride) {
    const uint32_t draw = i / PIECES, piece = i % PIECES;
    uint4 v = host_src[i];
    if (draw < n_draws && piece >= 4u && piece < 8u) {
      const uint4 col = host_src[draw * PIECES + piece - 4u];
      const uint32_t flags = reinterpret_cast<const uint32_t*>(host_src + draw * PIECES)[offsetof(DrawDesc, flags) / 4u];
      v = col;
      if (((flags >> F_KIND_SHIFT) & 3u) == PIPE_MESH) {
        float out[4];
        matvec4(scene.viewproj, u2f(col.x), u2f(col.y), u2f(col.z), u2f(col.w), out);  // C0
        v = make_uint4(f2u(out[0]), f2u(out[1]), f2u(out[2]), f2u(out[3]));
      }
    }
    dst[i] = v;
  }
  for (uint32_t i = t; i < n_zero; i += stride) zero[i] = make_uint4(0, 0, 0, 0);
}

void launch_prologue(const void* host_src, void* dst, size_t copy_bytes, void* zero, size_t zero_bytes, uint32_t n_draws,
                     const SvrSceneData& scene, hipStream_t s) {
  uint32_t n_copy = (uint32_t)((copy_bytes + 15) / 16), n_zero = (uint32_t)((zero_bytes + 15) / 16);
  uint32_t blocks = std::min<uint32_t>(256u, (std::max(n_copy, n_zero) + 255u) / 256u);
  hipLaunchKernelGGL(prologue_kernel, dim3(std::max(blocks, 1u)), dim3(256), 0, s, (const uint4*)host_src, (uint4*)dst, n_copy,
                     (uint4*)zero, n_zero, n_draws, scene);
}
void launch_mesh_vert(const SvrVertex* vtx, uint32_t first, uint32_t n, const float* world16,
                      const float* viewproj16, const float* color_factors4, float* out_clip,
                      float* out_varyings, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(mesh_vert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, vtx, first, n, world16, viewproj16,
                     color_factors4, out_clip, out_varyings);
}

void launch_vertex_shader(const SvrVertex* vtx, uint32_t first, uint32_t n, const float* matrix16, float* out_clip,
                          float* out_varyings, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(vertex_shader_kernel, dim3((n + 255) / 256), dim3(256), 0, s, vtx, first, n, matrix16, out_clip, out_varyings);
}

}  // namespace svr
