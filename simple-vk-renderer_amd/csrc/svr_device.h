// svr_device.h — device-side data layout and the arithmetic contract as
// __device__ functions, shared by the geometry / binning / tile kernels.  gfx950 only.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/svr.h"

namespace svr {

constexpr int TILE = 32;           // tile edge in pixels; one 256-thread workgroup per tile
constexpr int TILE_SHIFT = 5;
constexpr int WAVE = 64;
constexpr float GUARD = 16384.0f;  // guard band in pixels (C3)
constexpr uint32_t NO_REC = 0xffffffffu;
#ifndef SVR_RCP_VARIANT
#define SVR_RCP_VARIANT 1
#endif

enum PipelineKind : uint32_t { PIPE_MESH = 0, PIPE_COLORED_TRIANGLE = 1, PIPE_TEX_IMAGE = 2 };

// flags of DrawDesc / TriRec
constexpr uint32_t F_T1 = 1u;            // edge 1 is not top-left (biased by -1)
constexpr uint32_t F_T2 = 2u;            // edge 2 is not top-left
constexpr uint32_t F_KIND_SHIFT = 4;     // 2 bits
constexpr uint32_t F_TRANSPARENT = 256u;

// SVR_OPT_TUNING bits: switch an optimisation off at run time so it can be A/B-timed in one process
constexpr uint32_t TUNE_NO_TILE_ORDER = 1u;   // tile kernel walks tiles row-major instead of heaviest-first
constexpr uint32_t TUNE_NO_LAZY_CLEAR = 4u;   // svr_clear_color runs its own kernel at once instead of riding in the next pass
constexpr uint32_t TUNE_NO_PIPELINE = 2u;     // geometry+binning on the caller's stream too (no overlap between passes)
constexpr uint32_t TUNE_NO_SPLIT = 8u;        // heavy tiles are not cut into four row quarters
constexpr uint32_t TUNE_HIZ = 64u;            // ... with it whatever the pass's bins hold (tests; the default is a choice per pass: k_tile.hip HIZ_AVG_ENTRIES)
constexpr uint32_t TUNE_NO_HIZ = 32u;         // phase A of the tile kernel walks every triangle (no hierarchical depth test)
constexpr uint32_t TUNE_NO_POLL = 16u;        // finished passes are validated at fences only (tests: an overflow is always found late)

// A heavy tile is rendered by FOUR workgroups, one per 8 rows (tile kernel "quarters").  The slowest tile
// bounds the tile kernel however many CUs are idle, and ordered blending makes a tile with a deep transparent
// bin the slowest by far (a curtain seen edge-on: 1270 triangles in a 1080p tile, 400K of its 495K cycles,
// where the chip's mean load per workgroup slot is 93K) — all the more when the frame is cut into the row
// bands of a multi-GPU run.  1920x1080: 0.214 -> 0.141 ms; one eighth of a 4K frame: 0.150 -> 0.092 ms.
// A quarter repeats the tile's per-triangle staging and its sort, so splitting costs total work: a tile is
// split only when its estimated cost (tile_cost: thousands of cycles, fitted to SVR_OPT_TILE_CYCLES
// readings) exceeds SPLIT_MIN_COST and the pass's mean load per workgroup slot.  The four quarters of every
// split tile head the launch (blockIdx < SPLIT_EXTRA; unclaimed slots return at once).
constexpr uint32_t SPLIT_MAX = 256;            // tiles split per pass at most (first come, first served)
constexpr uint32_t SPLIT_EXTRA = 4 * SPLIT_MAX;
constexpr uint32_t SPLIT_MIN_COST = 100;
constexpr uint32_t SPLIT_TILES_MAX = 4096;     // passes over more tiles run the tile kernel built without the quarter path
constexpr uint32_t ROW_COST_MAX = 512;        // tile rows of the largest target (16384 / 32)
#ifndef SVR_AB_TILE_SLOTS  // A/B builds only (tools/build_variant.sh)
#define SVR_AB_TILE_SLOTS 1024
#endif
constexpr uint32_t TILE_SLOTS = SVR_AB_TILE_SLOTS;  // resident tile workgroups the split rule's mean load is taken over
__host__ __device__ inline uint32_t tile_cost(uint32_t n_op, uint32_t n_tr) { return 40u + (n_op >> 3) + n_tr - (n_tr >> 2); }
constexpr uint32_t SPLIT_SORT_MAX = 1392;      // quarters sort in LDS, out of place (k_tile.hip SORT_CAP): larger transparent bins stay whole

// One draw call (RenderObject after cull+sort), 192 bytes.  Read by the setup kernel through scalar loads
// (the draw is wave-uniform): the whole record arrives in one round trip.
struct DrawDesc {
  float mat[16];            // MESH: world matrix (push constant); TEX_IMAGE: render_matrix
  float mvp[16];            // MESH: sceneData.viewproj * mat, once per draw (the C0 fma chain, host or k_flatten); else = mat
  float color_factors[4];   // materialData.color_factors
  const SvrVertex* vtx;     // vertex_buf_address
  const uint32_t* idx;      // index buffer + first_index
  uint32_t tri_count;
  uint32_t tri_base;        // sequence number of the draw's first triangle (submission order)
  uint32_t tex;             // TexBinding index
  uint32_t flags;           // kind << F_KIND_SHIFT | F_TRANSPARENT
  const float* groups;      // the mesh's index-group table (svr_upload_mesh): GROUP_WORDS words per group = float[6] min xyz, max xyz of the vertices
                            // named by indices [192 g, 192 g + 192); NULL = none (the wave chunks are never skipped)
  uint32_t first_index;     // the draw's offset in the mesh's index buffer: locates its chunks' groups
  uint32_t pad;
};
static_assert(sizeof(DrawDesc) == 192 && offsetof(DrawDesc, vtx) == 144 && offsetof(DrawDesc, groups) == 176, "DrawDesc layout");
constexpr uint32_t GROUP_INDICES = 192;  // 64 triangles
// A draw's triangles are cut into wave chunks ALONG the grid of its mesh's index groups (triangle number
// first_index / 3 + t, in steps of 64), not from the draw's first triangle on: a chunk then lies inside ONE group — one box
// to cull it with, one short run of the vertex buffer to shade (chunks that straddled two groups were culled by the union
// of two boxes and shaded the union of two runs, often past the 128 vertices the staging holds).  The first and last
// chunk of a draw may be short.  A draw whose first index is not a multiple of 3 keeps the plain cut (phase 0).
__host__ __device__ inline uint32_t chunk_phase(uint32_t first_index) { return first_index % 3u == 0u ? (first_index / 3u) & 63u : 0u; }
__host__ __device__ inline uint32_t chunk_count(uint32_t first_index, uint32_t tri_count) {
  return tri_count ? (chunk_phase(first_index) + tri_count + 63u) / 64u : 0u;
}
__host__ __device__ inline uint32_t chunk_first(uint32_t first_index, uint32_t k) { return k ? 64u * k - chunk_phase(first_index) : 0u; }
__host__ __device__ inline uint32_t chunk_len(uint32_t first_index, uint32_t tri_count, uint32_t first_tri) {
  const uint32_t to_grid = 64u - ((chunk_phase(first_index) + first_tri) & 63u), left = tri_count - first_tri;
  return to_grid < left ? to_grid : left;
}
constexpr uint32_t GROUP_WORDS = 8;      // a group's entry: float[6] box of the vertices its indices name, then their lowest and highest vertex index (uint32)

// 64 consecutive triangles of one draw: the unit of work of one wave of the setup kernel.
struct WaveChunk {
  uint32_t draw;
  uint32_t first_tri;
};

// Texture + sampler as the fragment stage sees them (combined image sampler).  24 meaningful bytes
// so the setup kernel can copy it into every TriRec: shading then needs no descriptor fetch.
// Mip level l of an image lives at byte offset mip_offset(lw, lh, l) from base (levels are laid
// out as if the image were padded to 2^lw x 2^lh, so the offset is a closed form), row pitch =
// the level's true width max(w >> l, 1).
struct TexBinding {
  uint32_t base_off;        // RGBA8 texels of level 0: byte offset in the context's texel arena (FrameParams::tex_arena)
  uint32_t pad0;
  uint32_t wh;              // w | h << 16   (extent <= 16384 each)
  uint32_t info;            // lw | lh << 8 | levels << 16 | filters << 24; filters = mag | min<<1 | mip<<2
  float min_lod, max_lod;
  uint32_t pad[2];
};
static_assert(sizeof(TexBinding) == 32, "TexBinding layout");

// = sum over k < level of 4 * 2^max(lw-k,0) * 2^max(lh-k,0), for level <= max(lw, lh), without a loop
// (the fragment stage evaluates it twice per pixel).  While both extents halve the sum is
// 2^(lw+lh+4-2l) * (4^l - 1)/3, and (4^l - 1)/3 is the bit pattern 0101..01 with l ones; past the smaller
// extent only the larger one halves: a geometric tail of 4 * 2^(b-k).
__host__ __device__ inline uint32_t mip_offset(uint32_t lw, uint32_t lh, uint32_t level) {
  const uint32_t a = lw < lh ? lw : lh, b = lw < lh ? lh : lw;
  const uint32_t l1 = level < a ? level : a;
  const uint32_t ones = l1 ? (0x55555555u >> (32u - 2u * l1)) : 0u;
  uint32_t off = ones << ((lw + lh + 4u - 2u * l1) & 31u);
  if (level > a) off += 4u * ((1u << (b - a + 1u)) - (1u << (b - level + 1u)));
  return off;
}

// A set-up triangle, 256 bytes: first half is all the coverage/depth loop reads, second half only
// the winners' shading reads.  Edge functions are evaluated at integer pixel indices (px,py):
// e_i = A[i]*px + B[i]*py + C[i]  (C already carries the top-left bias); everything is an exact
// integer < 2^53 held in a double.  (An int32 form for triangles under 64 px, evaluated with
// v_mad_i32_i24, was measured 7 % SLOWER in the tile kernel: at its occupancy a lone wave issues one
// VALU instruction per ~4 cycles, which is also the cost of a 16-lane/clk v_fma_f64.)
struct TriRec {
  int16_t minx, miny, maxx, maxy;  // inclusive pixel bbox clamped to the scissor; minx>maxx = invalid
  uint32_t key;                    // make_key(): (submission sequence number + 1) << 2 | common-case shading << 1 | came-through-the-clipper
  uint32_t flags;
  float z0, dz1, dz2, inv_area;
  double A[3], B[3], C[3];
  uint32_t tex_off, tex_pad;       // the draw's TexBinding, copied in (offset 104): arena offset of level 0
  uint32_t tex_wh, tex_info;
  float tex_min_lod, tex_max_lod;
  // ---- shading half
  float q0, dq1, dq2;              // 1/w
  float a0[8], da1[8], da2[8];     // varyings pre-divided by w: normal.xyz, color.rgb, uv
  float pad2[5];
};
static_assert(sizeof(TriRec) == 256, "TriRec layout");
static_assert(offsetof(TriRec, A) == 32 && offsetof(TriRec, tex_off) == 104 && offsetof(TriRec, tex_wh) == 112 &&
                  offsetof(TriRec, q0) == 128, "TriRec layout");

struct ClipItem {
  uint32_t draw;
  uint32_t tri;
};

// device-side frame counters (zeroed by a memset node at the head of every pass)
struct Counters {
  uint32_t n_clip;         // triangles queued for the clipper
  uint32_t n_extra;        // records appended by the clipper
  uint32_t total_entries;  // bin entries (written by the scan)
  uint32_t overflow;       // bit0: clip queue, bit1: extra records, bit2: bin entries
  unsigned long long rasterized;
  unsigned long long shaded;
  unsigned long long binned;
  uint32_t n_big;          // triangles over 16 tiles queued by the setup kernel (binned wave-per-record)
  uint32_t n_pairs;        // (bin, record) pairs the setup kernel appended (keeps counting past pair capacity)
  uint32_t n_pairs_rest;   // pairs the count launch appends itself (big triangles, clipper pieces), from the list's END downwards
  uint32_t sort_used;      // words of the sort arena handed out to transparent bins too large for an LDS sort
  uint32_t n_split;        // tiles fill_kernel cut into quarters (keeps counting past SPLIT_MAX)
  uint32_t cost_sum;       // sum of tile_cost over the pass's tiles (offsets_kernel)
  // device flatten (k_flatten.hip): what the host only knows upper bounds of
  uint32_t flat_draws;     // draws after culling (visible opaque + transparent)
  uint32_t flat_tris;      // their triangles
  uint32_t flat_chunks;    // their wave chunks: the setup kernel's real grid
  uint32_t flat_culled;    // opaque objects rejected by is_visible
  uint32_t hiz_bad;        // instrumented passes: fragments the hierarchical depth test would have dropped although they win (must stay 0)
  uint32_t pad1;
  unsigned long long pad2;  // 96 bytes: the counters head the tile-counter allocation (zeroed by the prologue)
};
static_assert(sizeof(Counters) == 96, "Counters layout");

// resource tables of the device flatten pass (handle - 1 indexes them)
struct MeshEntry {
  const SvrVertex* vtx;
  const uint32_t* idx;
  const float* groups;
  uint64_t pad;
};
struct MatEntry {
  float cf[4];
  uint32_t pass;
  uint32_t pad[3];
};

struct FrameParams {
  // targets
  void* color;
  float* depth;
  uint32_t W, H;                  // target extent == viewport
  uint32_t sx, sy, sw, sh;        // scissor
  uint32_t tiles_x, tiles_y, n_tiles;
  uint32_t rstride, roff;         // svr_set_row_interleave: of the scissor's 32-row tile rows this pass owns those with
                                  // index % rstride == roff; tiles_y, bins and row costs count the owned rows only
  // geometry
  const DrawDesc* draws;
  const WaveChunk* chunks;
  uint32_t n_chunks;
  uint32_t n_tris;                // main record slots
  TriRec* recs;                   // [n_tris + extra_cap]
  uint32_t extra_cap;
  ClipItem* clip_queue;
  uint32_t clip_cap;
  uint32_t* big_queue;            // [n_tris] main records whose bbox spans more than 16 tiles
  // bins
  uint32_t* tile_count;           // [2*n_tiles]: opaque bins then transparent bins
  uint32_t* tile_offset;          // [2*n_tiles]
  uint32_t* cls_count;            // [80] zeroed per pass: [0,33) tiles per weight class, [40,73) placement cursors
  uint32_t* row_cost;             // [ROW_COST_MAX] zeroed per pass: sum of tile_cost over every tile row (offsets_kernel)
  uint32_t* host_row_cost;        // pinned: the tile kernel posts row_cost there (svr_get_row_costs: load-balanced row bands)
  Counters* host_counters;        // pinned host copy, written by report_kernel behind the tile kernel
  uint32_t* host_failed_seq;      // pinned: op_seq of the first pass that overflowed since the last recovery (0 = none)
  uint32_t op_seq;                // this pass's number in the context's operation log (never 0)
  uint32_t lazy_clear;            // 1: pixels this pass does not cover get the clear value clear_lo/clear_hi (deferred svr_clear_color)
  uint32_t clear_lo, clear_hi;    // the encoded texel (RGBA16F: 4 halves; RGBA8: clear_lo)
  uint32_t flatten;               // 1: draws/chunks were built on the device; n_tris / n_chunks are upper bounds
  uint32_t* poison;               // sticky per-context flag: an earlier pass overflowed, target writes are void
  uint2* pairs;                   // [bin_cap] (bin, record): what binning scatters, in emission order
  uint32_t* pair_slot;            // [bin_cap] position of the pair inside its bin
  uint4* tile_info;               // [2*n_tiles] launch slot -> {tile, n_opaque, offset_opaque, n_transparent}, {offset_transparent, sort_base,-,-}
  unsigned long long* sort_arena; // [sort_cap] (key << 32 | record) scratch of transparent bins over the LDS sort capacity
  uint32_t sort_cap;
  uint32_t* tile_order;           // [n_tiles] launch order of the tile kernel, heaviest first
  uint32_t* bins;
  uint32_t bin_cap;
  Counters* counters;
  const TexBinding* tex;
  const uint8_t* tex_arena;       // every image of the context: texel addresses are this + a 32-bit offset
  uint32_t instrument;            // count fragments/triangles with device atomics (not in timed runs)
  int trace_x, trace_y;           // instrumented passes only: dump the shading of this pixel
  float* trace_buf;               // 64 floats or NULL
  uint32_t* tile_cycles;          // SVR_OPT_TILE_CYCLES: [n_tiles][4] shader-clock cycles of phases A..D
  uint32_t tuning;                // SVR_OPT_TUNING bits (A/B switches for benchmarking, default 0)
  uint32_t pad_t;
  SvrSceneData scene;
};

// k_flatten.hip: cull + sort + per-object draw records on the device
struct FlattenParams {
  const SvrRenderObject* objects;  // pinned host memory: opaque list, then the transparent list
  SvrRenderObject* objects_dev;    // the same, pulled into device memory by cull_kernel (coalesced, once) for the kernels behind it
  uint32_t n_opaque, n_transparent;
  float viewproj[16];
  const MeshEntry* meshes;
  const MatEntry* materials;
  unsigned long long* keys;  // [n_opaque]
  uint32_t* draw_tris;       // [n_opaque + n_transparent]
  uint32_t* chunk_base;      // [n_opaque + n_transparent]
  DrawDesc* draws;
  WaveChunk* chunks;
  Counters* counters;
};

// ------------------------------------------------------------------------------------------------
// The tile rows a pass owns (svr_set_row_interleave): first pixel row of its local tile row `ty` ...
__device__ __forceinline__ int tile_row_y(const FrameParams& P, int ty) {
  return (int)P.sy + (ty * (int)P.rstride + (int)P.roff) * TILE;
}
// ... and the local tile rows [l0, l1] that pixel rows [miny, maxy] (inside the scissor) meet; none: l0 > l1
__device__ __forceinline__ void local_tile_rows(const FrameParams& P, int miny, int maxy, int& l0, int& l1) {
  const int g0 = (miny - (int)P.sy) >> TILE_SHIFT, g1 = (maxy - (int)P.sy) >> TILE_SHIFT;
  l0 = g0;
  l1 = g1;
  if (P.rstride > 1u) {  // pass-uniform
    const int s = (int)P.rstride, b = g1 - (int)P.roff;
    l0 = (g0 - (int)P.roff + s - 1) / s;  // ceil; the dividend is never negative (roff < rstride)
    l1 = b < 0 ? -1 : b / s;
  }
}

__device__ __forceinline__ float u2f(uint32_t u) { return __uint_as_float(u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __float_as_uint(f); }

// 1.0f / x, correctly rounded (the contract's "IEEE 1/x"), without the compiler's division expansion
// (v_div_scale x2, v_rcp, five fma, v_div_fmas, v_div_fixup: ~11 instructions where the fragment stage runs
// three of them per pixel).  v_rcp_f32 is good to 1 ulp; VARIANT picks the refinement, and
// svr_debug_rcp_sweep (tests/test_parity_gpu.py::test_reciprocal_all_inputs) compares the result with the
// compiler's 1.0f / x over all 2^32 bit patterns.  Outside the exponent window in which neither the input,
// the result nor the residuals can be subnormal, and for the few significands the refinement gets wrong
// (found by that sweep), the division proper runs — a wave-uniform branch that scenes never take.
//   VARIANT 1: one Newton step   y + y*(1 - x*y)
//   VARIANT 2: two Newton steps
//   VARIANT 0: the production form (svr_device.h rcp_ieee)
template <int VARIANT>
__device__ __forceinline__ float rcp_refined(float x) {
  float y = __builtin_amdgcn_rcpf(x);
  float r = fmaf(-x, y, 1.0f);
  y = fmaf(r, y, y);
  if (VARIANT == 2) {
    r = fmaf(-x, y, 1.0f);
    y = fmaf(r, y, y);
  }
  return y;
}
__device__ __forceinline__ bool rcp_fast_ok(float x) {
  // biased exponent in [32, 222]: |x| in [2^-95, 2^96), so 1/x, x*y and the residual are all normal
  return ((f2u(x) >> 23) & 0xffu) - 32u <= 190u;
}
__device__ __forceinline__ float rcp_ieee(float x) {
#ifdef SVR_AB_PLAIN_DIV  // A/B builds only (tools/build_variant.sh): the compiler's division everywhere
  return 1.0f / x;
#endif
  if (__builtin_expect(!rcp_fast_ok(x), 0)) return 1.0f / x;
  return rcp_refined<SVR_RCP_VARIANT>(x);
}

// three at once (the fragment stage's 1/q at the pixel and at its two quad partners): one window test, one branch
__device__ __forceinline__ void rcp3_ieee(float a, float b, float c, float& ra, float& rb, float& rc) {
#ifdef SVR_AB_PLAIN_DIV
  const bool window = false;
#else
  const bool window = rcp_fast_ok(a) && rcp_fast_ok(b) && rcp_fast_ok(c);
#endif
  if (__builtin_expect(!window, 0)) {
    ra = 1.0f / a;
    rb = 1.0f / b;
    rc = 1.0f / c;
    return;
  }
  ra = rcp_refined<SVR_RCP_VARIANT>(a);
  rb = rcp_refined<SVR_RCP_VARIANT>(b);
  rc = rcp_refined<SVR_RCP_VARIANT>(c);
}

// post-vertex-shader vertex: gl_Position + 8 varying floats
struct VOut {
  float clip[4];
  float attr[8];
};

// C0/C1: column-major mat4 * vec4 as an fma chain over columns
__device__ __forceinline__ void matvec4(const float* m, float x, float y, float z, float w, float* out) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    float acc = m[0 + r] * x;
    acc = fmaf(m[4 + r], y, acc);
    acc = fmaf(m[8 + r], z, acc);
    acc = fmaf(m[12 + r], w, acc);
    out[r] = acc;
  }
}
__device__ __forceinline__ void matmul4(const float* a, const float* b, float* out) {
#pragma unroll
  for (int j = 0; j < 4; j++) matvec4(a, b[4 * j + 0], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3], out + 4 * j);
}

// 48-byte interleaved Vertex (src/vk_types.h:97-103) as three 16-byte loads
struct VertexRaw {
  float4 a, b, c;  // a = position.xyz, uv_x; b = normal.xyz, uv_y; c = color
};
__device__ __forceinline__ VertexRaw load_vertex(const SvrVertex* base, uint32_t index) {
  const float4* p = reinterpret_cast<const float4*>(base + index);
  VertexRaw v;
  v.a = p[0];
  v.b = p[1];
  v.c = p[2];
  return v;
}

// shaders/mesh.vert:29-38
__device__ __forceinline__ void mesh_vert(const VertexRaw& v, const float* mvp, const float* world,
                                          const float* color_factors, VOut& o) {
  matvec4(mvp, v.a.x, v.a.y, v.a.z, 1.0f, o.clip);
#pragma unroll
  for (int r = 0; r < 3; r++) {
    float acc = world[0 + r] * v.b.x;
    acc = fmaf(world[4 + r], v.b.y, acc);
    acc = fmaf(world[8 + r], v.b.z, acc);
    o.attr[r] = acc;
  }
  o.attr[3] = v.c.x * color_factors[0];
  o.attr[4] = v.c.y * color_factors[1];
  o.attr[5] = v.c.z * color_factors[2];
  o.attr[6] = v.a.w;
  o.attr[7] = v.b.w;
}

// shaders/colored_triangle_mesh.vert:28-38
__device__ __forceinline__ void colored_triangle_mesh_vert(const VertexRaw& v, const float* render_matrix, VOut& o) {
  matvec4(render_matrix, v.a.x, v.a.y, v.a.z, 1.0f, o.clip);
  o.attr[0] = o.attr[1] = o.attr[2] = 0.0f;
  o.attr[3] = v.c.x;
  o.attr[4] = v.c.y;
  o.attr[5] = v.c.z;
  o.attr[6] = v.a.w;
  o.attr[7] = v.b.w;
}

// shaders/colored_triangle.vert:6-25
__device__ __forceinline__ void colored_triangle_vert(int i, VOut& o) {
  o.clip[0] = (i == 0) ? 1.0f : (i == 1 ? -1.0f : 0.0f);
  o.clip[1] = (i == 2) ? -1.0f : 1.0f;
  o.clip[2] = 0.0f;
  o.clip[3] = 1.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) o.attr[k] = 0.0f;
  o.attr[3] = (i == 0) ? 1.0f : 0.0f;
  o.attr[4] = (i == 1) ? 1.0f : 0.0f;
  o.attr[5] = (i == 2) ? 1.0f : 0.0f;
}

// C2: outcodes against the Vulkan clip volume
enum { OC_NEAR = 1, OC_FAR = 2, OC_L = 4, OC_R = 8, OC_T = 16, OC_B = 32 };
__device__ __forceinline__ int outcode(const float* c) {
  int oc = 0;
  if (c[2] > c[3]) oc |= OC_NEAR;
  if (c[2] < 0.0f) oc |= OC_FAR;
  if (c[0] < -c[3]) oc |= OC_L;
  if (c[0] > c[3]) oc |= OC_R;
  if (c[1] < -c[3]) oc |= OC_T;
  if (c[1] > c[3]) oc |= OC_B;
  return oc;
}

struct ScreenV {
  float xs, ys, zs, rw;
  bool ok;
};
// C3: perspective divide + viewport (0,0,W,H,0,1)
__device__ __forceinline__ ScreenV to_screen(const float* c, float hw, float hh) {
  ScreenV s;
  s.rw = rcp_ieee(c[3]);
  s.xs = fmaf(c[0] * s.rw, hw, hw);
  s.ys = fmaf(c[1] * s.rw, hh, hh);
  s.zs = c[2] * s.rw;
  s.ok = (fabsf(s.xs) <= GUARD) && (fabsf(s.ys) <= GUARD);
  return s;
}

__device__ __forceinline__ void store_invalid(TriRec* rec) {
  uint4 h;
  h.x = 1u;           // minx = 1, miny = 0
  h.y = 0u;           // maxx = 0, maxy = 0
  h.z = 0u;
  h.w = 0u;
  *reinterpret_cast<uint4*>(rec) = h;
}

// C4..C6: snap, orient, edge functions, attribute deltas.  Returns false when the triangle is
// dropped (zero area or no pixel centre inside the scissor).
// The per-triangle key the tile kernel resolves visibility and order with.  Two flag bits ride below the
// submission number (they cannot change an order between different triangles):
//   bit 0  the record came through the clipper: the main slot only links to the pieces
//   bit 1  the fragment stage is the common case — mesh.frag, a LINEAR/LINEAR/MIPMAP_LINEAR sampler, an image
//          with power-of-two extents —
//          for which the tile kernel has a specialised instance (a wave whose pixels all carry it takes it)
__device__ __forceinline__ uint32_t make_key(uint32_t seq, uint32_t draw_flags, const TexBinding& tex, bool clipped) {
  bool pow2 = (tex.wh & 0xffffu) == (1u << (tex.info & 0xffu)) && (tex.wh >> 16) == (1u << ((tex.info >> 8) & 0xffu));
  bool common = ((draw_flags >> F_KIND_SHIFT) & 3u) == PIPE_MESH && (tex.info >> 24) == 7u && pow2;
  return ((seq + 1u) << 2) | (common ? 2u : 0u) | (clipped ? 1u : 0u);
}

// what binning needs of a set-up triangle, still in registers
struct TriGeom {
  int minx, miny, maxx, maxy;
  double A[3], B[3], C[3];
};

__device__ inline bool setup_triangle(const FrameParams& P, const VOut* v0, const VOut* v1, const VOut* v2,
                                      ScreenV s0, ScreenV s1, ScreenV s2, uint32_t key, uint32_t draw_flags,
                                      const TexBinding& tex, uint4 (&rec)[16], TriGeom* geom) {
  int X0 = __float2int_rn(s0.xs * 256.0f), Y0 = __float2int_rn(s0.ys * 256.0f);
  int X1 = __float2int_rn(s1.xs * 256.0f), Y1 = __float2int_rn(s1.ys * 256.0f);
  int X2 = __float2int_rn(s2.xs * 256.0f), Y2 = __float2int_rn(s2.ys * 256.0f);
  long long area2 = (long long)(X1 - X0) * (long long)(Y2 - Y0) - (long long)(X2 - X0) * (long long)(Y1 - Y0);
  if (area2 == 0) return false;
  // negative orientation: vertices 1 and 2 trade places.  By value (selects): trading the POINTERS made the
  // three VOuts addressable, i.e. arrays in scratch memory written and re-read through the vector memory path
  const bool flip = area2 < 0;
  if (flip) {
    int t;
    t = X1; X1 = X2; X2 = t;
    t = Y1; Y1 = Y2; Y2 = t;
    area2 = -area2;
  }
  const float zs1 = flip ? s2.zs : s1.zs, zs2 = flip ? s1.zs : s2.zs;
  const float rw1 = flip ? s2.rw : s1.rw, rw2 = flip ? s1.rw : s2.rw;
  int xmin = min(X0, min(X1, X2)), xmax = max(X0, max(X1, X2));
  int ymin = min(Y0, min(Y1, Y2)), ymax = max(Y0, max(Y1, Y2));
  int pminx = (xmin + 127) >> 8, pmaxx = (xmax - 128) >> 8;
  int pminy = (ymin + 127) >> 8, pmaxy = (ymax - 128) >> 8;
  pminx = max(pminx, (int)P.sx);
  pminy = max(pminy, (int)P.sy);
  pmaxx = min(pmaxx, (int)(P.sx + P.sw) - 1);
  pmaxy = min(pmaxy, (int)(P.sy + P.sh) - 1);
  if (pminx > pmaxx || pminy > pmaxy) return false;
  if (P.rstride > 1u) {  // a rank of the interleaved multi-GPU form: triangles that meet none of its tile rows go no further
    int l0, l1;
    local_tile_rows(P, pminy, pmaxy, l0, l1);
    if (l0 > l1) return false;
  }

  int Xs[3] = {X0, X1, X2}, Ys[3] = {Y0, Y1, Y2};
  double A[3], B[3], C[3];
  uint32_t flags = draw_flags;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    int a = (i + 1) % 3, b = (i + 2) % 3;
    long long dx = (long long)Xs[b] - Xs[a], dy = (long long)Ys[b] - Ys[a];
    long long Ae = -dy, Be = dx, Ce = -dx * Ys[a] + dy * Xs[a];
    bool top_left = (dy < 0) || (dy == 0 && dx > 0);
    long long Cu = Ce + 128 * (Ae + Be);
    A[i] = (double)(Ae * 256);
    B[i] = (double)(Be * 256);
    C[i] = (double)(Cu + (top_left ? 0 : -1));
    if (!top_left && i == 1) flags |= F_T1;
    if (!top_left && i == 2) flags |= F_T2;
  }
  float inv_area = rcp_ieee((float)(double)area2);
  if (geom) {
    geom->minx = pminx; geom->miny = pminy; geom->maxx = pmaxx; geom->maxy = pmaxy;
#pragma unroll
    for (int i = 0; i < 3; i++) {
      geom->A[i] = A[i]; geom->B[i] = B[i]; geom->C[i] = C[i];
    }
  }

  // the record, as its sixteen 16-byte pieces (struct TriRec); the caller stores it
  uint4 h;
  h.x = ((uint32_t)(uint16_t)pminx) | ((uint32_t)(uint16_t)pminy << 16);
  h.y = ((uint32_t)(uint16_t)pmaxx) | ((uint32_t)(uint16_t)pmaxy << 16);
  h.z = key;
  h.w = flags;
  rec[0] = h;
  rec[1] = make_uint4(f2u(s0.zs), f2u(zs1 - s0.zs), f2u(zs2 - s0.zs), f2u(inv_area));
  auto pack2 = [](double a, double b) {
    unsigned long long ua = (unsigned long long)__double_as_longlong(a), ub = (unsigned long long)__double_as_longlong(b);
    return make_uint4((uint32_t)ua, (uint32_t)(ua >> 32), (uint32_t)ub, (uint32_t)(ub >> 32));
  };
  rec[2] = pack2(A[0], A[1]);
  rec[3] = pack2(A[2], B[0]);
  rec[4] = pack2(B[1], B[2]);
  rec[5] = pack2(C[0], C[1]);
  {
    unsigned long long uc = (unsigned long long)__double_as_longlong(C[2]);
    rec[6] = make_uint4((uint32_t)uc, (uint32_t)(uc >> 32), tex.base_off, 0u);
  }
  rec[7] = make_uint4(tex.wh, tex.info, f2u(tex.min_lod), f2u(tex.max_lod));
  // shading half
  float rw0 = s0.rw;
  float sh[32];
  sh[0] = rw0;
  sh[1] = rw1 - rw0;
  sh[2] = rw2 - rw0;
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const float a1 = flip ? v2->attr[k] : v1->attr[k], a2 = flip ? v1->attr[k] : v2->attr[k];
    float p0 = v0->attr[k] * rw0, p1 = a1 * rw1, p2 = a2 * rw2;
    sh[3 + k] = p0;
    sh[11 + k] = p1 - p0;
    sh[19 + k] = p2 - p0;
  }
#pragma unroll
  for (int k = 27; k < 32; k++) sh[k] = 0.0f;
#pragma unroll
  for (int k = 0; k < 8; k++) rec[8 + k] = make_uint4(f2u(sh[4 * k]), f2u(sh[4 * k + 1]), f2u(sh[4 * k + 2]), f2u(sh[4 * k + 3]));
  return true;
}

}  // namespace svr
