// svr_oracle.cpp — CPU scalar oracle for the draw_geometry path.  TEST INFRASTRUCTURE ONLY.
//
// This file is the checker, never the product: only tests/, __graft_entry__.smoke() and
// bench.py's cpu_baseline leg may load oracle/libsvr_oracle.so.  The product
// (simple-vk-renderer_amd/csrc/libsvr_hip.so) shares no code with it.
//
// PARITY UNPINNED: the reference ships no tests, golden images or fixtures and
// cannot be built here (needs Vulkan, GLFW, glm, fastgltf, VMA, ImGui; submodule dirs are empty).
// The oracle is therefore a restatement of the reference's shaders and pipeline state plus the
// Vulkan rules for the fixed-function stages, pinned only by analytic known-answer tests
// (tests/test_oracle_kat.py).  (The one runnable piece of the reference, its vendored stb_image, pins the
// glTF loader's image decoders instead: oracle/ref_stb_decode.cpp, tests/make_golden_images.py.)
// Where Vulkan leaves arithmetic implementation-defined, DESIGN.md
// §"Arithmetic contract" (C0..C13) fixes one choice; the comments below cite those items.
//
// What is restated, with the reference file:line each part follows:
//   is_visible                 src/vk_engine.cpp:56-86
//   draw_geometry              src/vk_engine.cpp:1357-1477 (cull, sort, opaque then transparent)
//   pipeline state             src/vk_engine.cpp:1619-1688, src/vk_pipelines.cpp:126-207
//   attachments                src/vk_initializers.cpp:117-164 (colour LOAD, depth CLEAR 0.0)
//   mesh.vert / mesh.frag      shaders/mesh.vert:29-38, shaders/mesh.frag:12-19
//   colored_triangle.*         shaders/colored_triangle.vert:6-25, .frag:9-12
//   colored_triangle_mesh.vert shaders/colored_triangle_mesh.vert:28-38
//   tex_image.frag             shaders/tex_image.frag:10-12
//   generate_mipmaps           src/vk_images.cpp:66-133
//   sampler state              src/vk_loader.cpp:197-211, src/vk_engine.cpp:252-261
//
// Plain scalar C++17, no intrinsics; build with -O2 -ffp-contract=off (oracle/Makefile).
// fmaf() is used exactly where the contract says "fma".

#include "../include/svr.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

inline uint32_t f2u(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  return u;
}
inline float u2f(uint32_t u) {
  float f;
  std::memcpy(&f, &u, 4);
  return f;
}

// ---------------------------------------------------------------- fp16 <-> fp32 (RTE), C10
uint16_t f32_to_f16(float f) {
  uint32_t x = f2u(f);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) {  // inf / nan
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax >= 0x477ff000u) {  // >= 65520 rounds to inf
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax >= 0x38800000u) {  // normal half
    uint32_t mant = ax & 0x7fffffu;
    uint32_t exp = (ax >> 23) - 112;
    uint32_t h = (exp << 10) | (mant >> 13);
    uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
  }
  if (ax < 0x33000000u) return (uint16_t)sign;  // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
  // subnormal half
  uint32_t mant = (ax & 0x7fffffu) | 0x800000u;
  int shift = 126 - (int)(ax >> 23);  // 14..24
  uint32_t h = mant >> shift;
  uint32_t rem = mant & ((1u << shift) - 1u);
  uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (h & 1u))) h++;
  return (uint16_t)(sign | h);
}

float f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1fu;
  uint32_t mant = h & 0x3ffu;
  if (exp == 0) {
    if (mant == 0) return u2f(sign);
    // subnormal: value = mant * 2^-24
    float v = (float)mant * 0x1p-24f;
    return (sign ? -v : v);
  }
  if (exp == 31) return u2f(sign | 0x7f800000u | (mant << 13));
  return u2f(sign | ((exp + 112) << 23) | (mant << 13));
}

inline uint8_t f32_to_unorm8(float f) {  // clamp, *255, round-to-nearest-even
  float c = fminf(fmaxf(f, 0.0f), 1.0f);
  return (uint8_t)lrintf(c * 255.0f);
}
const float kInv255 = 0x1.010102p-8f;  // fp32 nearest to 1/255 (0x3B808081), C9
inline float unorm8_to_f32(uint8_t c) { return (float)c * kInv255; }

// ---------------------------------------------------------------- resources
struct Mesh {
  std::vector<uint32_t> idx;
  std::vector<SvrVertex> vtx;
  bool alive = false;
};
struct Image {
  uint32_t w = 0, h = 0, levels = 0;
  std::vector<std::vector<uint8_t>> mip;
  std::vector<uint32_t> lw, lh;
  bool alive = false;
};
struct Sampler {
  SvrSamplerDesc d;
};
struct Material {
  int pass;
  float color_factors[4];
  float metal_rough[4];
  uint32_t image, sampler;  // 0-based
};

enum PipelineKind { PIPE_MESH = 0, PIPE_COLORED_TRIANGLE = 1, PIPE_TEX_IMAGE = 2 };

// post-vertex-shader vertex: gl_Position + 8 varying floats
// MESH: n.xyz, color.rgb, uv.xy   (shaders/mesh.vert:8-10)
struct VOut {
  float clip[4];
  float attr[8];
};

struct SetupTri {
  int64_t A[3], B[3], C[3], Cu[3];  // edge i at pixel index (px,py): A*px+B*py+C (C biased, Cu not)
  float inv_area;
  float z0, dz1, dz2;
  float q0, dq1, dq2;  // 1/w
  float a0[8], da1[8], da2[8];
  int minx, miny, maxx, maxy;  // inclusive pixel bbox, already clamped to the scissor
  int kind;
  bool transparent;
  uint32_t key;  // submission sequence number + 1
  const Image* image;
  const Sampler* sampler;
};

struct DrawCmd {
  int kind;
  const Mesh* mesh;
  uint32_t first_index, index_count;
  float mat[16];  // MESH: world matrix; TEX_IMAGE: render_matrix
  const Material* material;
  const Image* image;
  const Sampler* sampler;
  bool transparent;
};

}  // namespace

struct SvrContext {
  uint32_t W = 0, H = 0;
  int color_format = SVR_COLOR_RGBA16F;
  std::vector<uint16_t> color16;  // RGBA16F
  std::vector<uint8_t> color8;    // RGBA8
  std::vector<float> depth;
  uint32_t sx = 0, sy = 0, sw = 0, sh = 0;
  uint32_t rstride = 1, roff = 0;      // svr_set_row_interleave
  uint32_t pass_rstride = 1;
  uint32_t* present_status = nullptr;  // svr_set_present_status (a host pointer here)
  uint32_t pass_sy = 0, pass_sh = 0;  // scissor rows of the last pass (svr_get_row_costs)
  std::vector<std::unique_ptr<Mesh>> meshes;
  std::vector<std::unique_ptr<Image>> images;
  std::vector<Sampler> samplers;
  std::vector<Material> materials;
  SvrStats stats{};
  int threads = 1;
  int trace_x = -1, trace_y = -1;
  float trace[64] = {0};
};

namespace {

// ---------------------------------------------------------------- mip generation, C13
// vkCmdBlitImage 2:1 LINEAR blit per level (src/vk_images.cpp:95-128), restated in exact integer
// arithmetic: source coordinate of destination texel centre i is ((2i+1)*sw - dw) / (2dw); bilinear
// weights are the exact rationals; the weighted sum is rounded to nearest, ties to even.
void downsample_level(const std::vector<uint8_t>& src, uint32_t sw, uint32_t sh,
                      std::vector<uint8_t>& dst, uint32_t dw, uint32_t dh) {
  dst.resize((size_t)dw * dh * 4);
  for (uint32_t j = 0; j < dh; j++) {
    int64_t ny = (int64_t)(2 * j + 1) * sh - dh;  // numerator over 2*dh
    int64_t dy = 2 * (int64_t)dh;
    int64_t j0 = ny >= 0 ? ny / dy : -((-ny + dy - 1) / dy);
    int64_t wy1 = ny - j0 * dy, wy0 = dy - wy1;
    int64_t j1 = j0 + 1;
    j0 = std::min<int64_t>(std::max<int64_t>(j0, 0), sh - 1);
    j1 = std::min<int64_t>(std::max<int64_t>(j1, 0), sh - 1);
    for (uint32_t i = 0; i < dw; i++) {
      int64_t nx = (int64_t)(2 * i + 1) * sw - dw;
      int64_t dx = 2 * (int64_t)dw;
      int64_t i0 = nx >= 0 ? nx / dx : -((-nx + dx - 1) / dx);
      int64_t wx1 = nx - i0 * dx, wx0 = dx - wx1;
      int64_t i1 = i0 + 1;
      i0 = std::min<int64_t>(std::max<int64_t>(i0, 0), sw - 1);
      i1 = std::min<int64_t>(std::max<int64_t>(i1, 0), sw - 1);
      int64_t den = dx * dy;
      for (int c = 0; c < 4; c++) {
        int64_t t00 = src[((size_t)j0 * sw + i0) * 4 + c], t10 = src[((size_t)j0 * sw + i1) * 4 + c];
        int64_t t01 = src[((size_t)j1 * sw + i0) * 4 + c], t11 = src[((size_t)j1 * sw + i1) * 4 + c];
        int64_t num = wy0 * (wx0 * t00 + wx1 * t10) + wy1 * (wx0 * t01 + wx1 * t11);
        int64_t q = num / den, r = num - q * den;
        if (2 * r > den || (2 * r == den && (q & 1))) q++;
        dst[((size_t)j * dw + i) * 4 + c] = (uint8_t)q;
      }
    }
  }
}

// ---------------------------------------------------------------- texture sampling, C8/C9
const float kG0 = 0x1.c51282p-2f, kG1 = -0x1.14a3a2p-2f, kG2 = 0x1.37536ap-3f, kG3 = -0x1.778474p-5f;

// lambda = 0.5*log2(rho2) with the contract's polynomial log2 (C8)
float lod_from_rho2(float rho2) {
  if (!(rho2 >= 0x1p-100f)) return -50.0f;
  if (!(rho2 <= 0x1p+100f)) return 50.0f;
  uint32_t bits = f2u(rho2);
  int e = (int)(bits >> 23) - 127;
  float m = u2f((bits & 0x7fffffu) | 0x3f800000u);
  float t = m - 1.0f;
  float g = fmaf(fmaf(fmaf(kG3, t, kG2), t, kG1), t, kG0);
  float s = t * (1.0f - t);
  float l = fmaf(s, g, t);
  return 0.5f * ((float)e + l);
}

inline void fetch_texel(const Image& img, uint32_t level, int i, int j, float out[4]) {
  const uint8_t* p = &img.mip[level][((size_t)j * img.lw[level] + i) * 4];
  for (int c = 0; c < 4; c++) out[c] = unorm8_to_f32(p[c]);
}

void sample_level(const Image& img, uint32_t level, int filter, float u, float v, float out[4]) {
  u = (fabsf(u) < 8388608.0f) ? u : 0.0f;
  v = (fabsf(v) < 8388608.0f) ? v : 0.0f;
  int wl = (int)img.lw[level], hl = (int)img.lh[level];
  float U = (u - floorf(u)) * (float)wl;
  float V = (v - floorf(v)) * (float)hl;
  if (filter == SVR_FILTER_NEAREST) {
    int i = (int)floorf(U), j = (int)floorf(V);
    if (i >= wl) i -= wl;
    if (j >= hl) j -= hl;
    fetch_texel(img, level, i, j, out);
    return;
  }
  float Uh = U - 0.5f, Vh = V - 0.5f;
  float fu = floorf(Uh), fv = floorf(Vh);
  float alpha = Uh - fu, beta = Vh - fv;
  int i0 = (int)fu, j0 = (int)fv;
  int i1 = i0 + 1, j1 = j0 + 1;
  if (i0 < 0) i0 += wl;
  if (i1 >= wl) i1 -= wl;
  if (j0 < 0) j0 += hl;
  if (j1 >= hl) j1 -= hl;
  float t00[4], t10[4], t01[4], t11[4];
  fetch_texel(img, level, i0, j0, t00);
  fetch_texel(img, level, i1, j0, t10);
  fetch_texel(img, level, i0, j1, t01);
  fetch_texel(img, level, i1, j1, t11);
  for (int c = 0; c < 4; c++) {
    float top = fmaf(alpha, t10[c] - t00[c], t00[c]);
    float bot = fmaf(alpha, t11[c] - t01[c], t01[c]);
    out[c] = fmaf(beta, bot - top, top);
  }
}

// texture(sampler2D, uv) with implicit LOD from the quad derivatives (C7..C9)
void sample_texture(const Image& img, const Sampler& smp, float u, float v, float dudx, float dvdx,
                    float dudy, float dvdy, float out[4]) {
  float W0 = (float)img.w, H0 = (float)img.h;
  float mx = dudx * W0, my = dvdx * H0;
  float nx = dudy * W0, ny = dvdy * H0;
  float rx2 = fmaf(mx, mx, my * my);
  float ry2 = fmaf(nx, nx, ny * ny);
  float rho2 = fmaxf(rx2, ry2);
  float lambda = lod_from_rho2(rho2);
  lambda = fminf(fmaxf(lambda, smp.d.min_lod), smp.d.max_lod);
  int filter = (lambda <= 0.0f) ? smp.d.mag_filter : smp.d.min_filter;
  int q = (int)img.levels - 1;
  if (smp.d.mipmap_mode == SVR_MIPMAP_NEAREST) {
    int d = (int)ceilf(lambda + 0.5f) - 1;
    d = std::min(std::max(d, 0), q);
    sample_level(img, (uint32_t)d, filter, u, v, out);
    return;
  }
  float lc = fminf(fmaxf(lambda, 0.0f), (float)q);
  float fl = floorf(lc);
  int dhi = (int)fl;
  float delta = lc - fl;
  float hi[4];
  sample_level(img, (uint32_t)dhi, filter, u, v, hi);
  if (delta == 0.0f) {
    for (int c = 0; c < 4; c++) out[c] = hi[c];
    return;
  }
  int dlo = std::min(dhi + 1, q);
  float lo[4];
  sample_level(img, (uint32_t)dlo, filter, u, v, lo);
  for (int c = 0; c < 4; c++) out[c] = fmaf(delta, lo[c] - hi[c], hi[c]);
}

// ---------------------------------------------------------------- vertex stage, C0/C1
// mat (column-major) times vec4 as an fma chain over columns
inline void matvec4(const float* m, float x, float y, float z, float w, float out[4]) {
  for (int r = 0; r < 4; r++) {
    float acc = m[0 + r] * x;
    acc = fmaf(m[4 + r], y, acc);
    acc = fmaf(m[8 + r], z, acc);
    acc = fmaf(m[12 + r], w, acc);
    out[r] = acc;
  }
}
// OpMatrixTimesMatrix: column j of the product = A * (column j of B)
inline void matmul4(const float* a, const float* b, float* out) {
  for (int j = 0; j < 4; j++) matvec4(a, b[4 * j + 0], b[4 * j + 1], b[4 * j + 2], b[4 * j + 3], out + 4 * j);
}

// shaders/mesh.vert:29-38
void mesh_vert(const SvrVertex& v, const float* mvp, const float* world, const float* color_factors,
               VOut& o) {
  matvec4(mvp, v.position[0], v.position[1], v.position[2], 1.0f, o.clip);
  for (int r = 0; r < 3; r++) {  // (world * vec4(n, 0)).xyz; the w=0 column adds exactly nothing
    float acc = world[0 + r] * v.normal[0];
    acc = fmaf(world[4 + r], v.normal[1], acc);
    acc = fmaf(world[8 + r], v.normal[2], acc);
    o.attr[r] = acc;
  }
  for (int c = 0; c < 3; c++) o.attr[3 + c] = v.color[c] * color_factors[c];
  o.attr[6] = v.uv_x;
  o.attr[7] = v.uv_y;
}

// shaders/colored_triangle_mesh.vert:28-38
void colored_triangle_mesh_vert(const SvrVertex& v, const float* render_matrix, VOut& o) {
  matvec4(render_matrix, v.position[0], v.position[1], v.position[2], 1.0f, o.clip);
  o.attr[0] = o.attr[1] = o.attr[2] = 0.0f;
  for (int c = 0; c < 3; c++) o.attr[3 + c] = v.color[c];
  o.attr[6] = v.uv_x;
  o.attr[7] = v.uv_y;
}

// shaders/colored_triangle.vert:6-25
void colored_triangle_vert(int vertex_index, VOut& o) {
  static const float positions[3][3] = {{1.f, 1.f, 0.f}, {-1.f, 1.f, 0.f}, {0.f, -1.f, 0.f}};
  static const float colors[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
  for (int c = 0; c < 3; c++) o.clip[c] = positions[vertex_index][c];
  o.clip[3] = 1.0f;
  for (int c = 0; c < 8; c++) o.attr[c] = 0.0f;
  for (int c = 0; c < 3; c++) o.attr[3 + c] = colors[vertex_index][c];
}

// ---------------------------------------------------------------- clipping, C2
enum { OC_NEAR = 1, OC_FAR = 2, OC_L = 4, OC_R = 8, OC_T = 16, OC_B = 32 };
inline int outcode(const float* c) {
  int oc = 0;
  if (c[2] > c[3]) oc |= OC_NEAR;
  if (c[2] < 0.0f) oc |= OC_FAR;
  if (c[0] < -c[3]) oc |= OC_L;
  if (c[0] > c[3]) oc |= OC_R;
  if (c[1] < -c[3]) oc |= OC_T;
  if (c[1] > c[3]) oc |= OC_B;
  return oc;
}
inline float plane_dist(int plane, const float* c) {
  switch (plane) {
    case 0: return c[3] - c[2];  // z <= w
    case 1: return c[2];         // z >= 0
    case 2: return c[3] + c[0];
    case 3: return c[3] - c[0];
    case 4: return c[3] + c[1];
    default: return c[3] - c[1];
  }
}
// Sutherland-Hodgman against the six planes of the Vulkan clip volume; a new vertex is always
// interpolated from the inside endpoint towards the outside endpoint so both triangles that share
// an edge produce the identical vertex.
int clip_polygon(VOut* poly, int n) {
  VOut tmp[12];
  for (int plane = 0; plane < 6 && n >= 3; plane++) {
    int m = 0;
    for (int i = 0; i < n; i++) {
      const VOut& a = poly[i];
      const VOut& b = poly[(i + 1) % n];
      float da = plane_dist(plane, a.clip), db = plane_dist(plane, b.clip);
      bool ina = da >= 0.0f, inb = db >= 0.0f;
      if (ina) tmp[m++] = a;
      if (ina != inb) {
        const VOut& pin = ina ? a : b;
        const VOut& pout = ina ? b : a;
        float din = ina ? da : db, dout = ina ? db : da;
        float t = din / (din - dout);
        VOut nv;
        for (int k = 0; k < 4; k++) nv.clip[k] = fmaf(t, pout.clip[k] - pin.clip[k], pin.clip[k]);
        for (int k = 0; k < 8; k++) nv.attr[k] = fmaf(t, pout.attr[k] - pin.attr[k], pin.attr[k]);
        tmp[m++] = nv;
      }
    }
    n = m;
    for (int i = 0; i < n; i++) poly[i] = tmp[i];
  }
  return n >= 3 ? n : 0;
}

struct ScreenV {
  float xs, ys, zs, rw;
  bool ok;
};
const float kGuard = 16384.0f;
// viewport transform, C3: viewport = (0,0,W,H,0,1)  (src/vk_engine.cpp:1421-1429)
inline ScreenV to_screen(const float* c, float hw, float hh) {
  ScreenV s;
  s.rw = 1.0f / c[3];
  s.xs = fmaf(c[0] * s.rw, hw, hw);
  s.ys = fmaf(c[1] * s.rw, hh, hh);
  s.zs = c[2] * s.rw;
  s.ok = (fabsf(s.xs) <= kGuard) && (fabsf(s.ys) <= kGuard);
  return s;
}

struct PassState {
  SvrContext* ctx;
  std::vector<SetupTri> tris;
  uint64_t binned = 0;
  uint32_t cur_key = 0;
};

// triangle setup, C4..C6
void emit_triangle(PassState& ps, const VOut* v0, const VOut* v1, const VOut* v2, ScreenV s0,
                   ScreenV s1, ScreenV s2, const DrawCmd& cmd) {
  SvrContext* ctx = ps.ctx;
  int64_t X[3], Y[3];
  const VOut* vv[3] = {v0, v1, v2};
  ScreenV ss[3] = {s0, s1, s2};
  for (int i = 0; i < 3; i++) {
    X[i] = (int64_t)lrintf(ss[i].xs * 256.0f);
    Y[i] = (int64_t)lrintf(ss[i].ys * 256.0f);
  }
  int64_t area2 = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0]);
  if (area2 == 0) return;
  if (area2 < 0) {
    std::swap(X[1], X[2]);
    std::swap(Y[1], Y[2]);
    std::swap(vv[1], vv[2]);
    std::swap(ss[1], ss[2]);
    area2 = -area2;
  }
  SetupTri t;
  int64_t xmin = std::min(X[0], std::min(X[1], X[2])), xmax = std::max(X[0], std::max(X[1], X[2]));
  int64_t ymin = std::min(Y[0], std::min(Y[1], Y[2])), ymax = std::max(Y[0], std::max(Y[1], Y[2]));
  int64_t pminx = (xmin + 127) >> 8, pmaxx = (xmax - 128) >> 8;
  int64_t pminy = (ymin + 127) >> 8, pmaxy = (ymax - 128) >> 8;
  pminx = std::max<int64_t>(pminx, ctx->sx);
  pminy = std::max<int64_t>(pminy, ctx->sy);
  pmaxx = std::min<int64_t>(pmaxx, (int64_t)ctx->sx + ctx->sw - 1);
  pmaxy = std::min<int64_t>(pmaxy, (int64_t)ctx->sy + ctx->sh - 1);
  if (pminx > pmaxx || pminy > pmaxy) return;
  t.minx = (int)pminx;
  t.maxx = (int)pmaxx;
  t.miny = (int)pminy;
  t.maxy = (int)pmaxy;
  // edge i is opposite vertex i: from vertex (i+1)%3 to vertex (i+2)%3
  for (int i = 0; i < 3; i++) {
    int a = (i + 1) % 3, b = (i + 2) % 3;
    int64_t dx = X[b] - X[a], dy = Y[b] - Y[a];
    // e(P) = dx*(P.y - Ya) - dy*(P.x - Xa), P = (256*px+128, 256*py+128)
    int64_t Ae = -dy, Be = dx, Ce = -dx * Y[a] + dy * X[a];
    bool top_left = (dy < 0) || (dy == 0 && dx > 0);
    t.A[i] = Ae * 256;
    t.B[i] = Be * 256;
    t.Cu[i] = Ce + 128 * (Ae + Be);
    t.C[i] = t.Cu[i] + (top_left ? 0 : -1);
  }
  t.inv_area = 1.0f / (float)area2;
  t.z0 = ss[0].zs;
  t.dz1 = ss[1].zs - ss[0].zs;
  t.dz2 = ss[2].zs - ss[0].zs;
  t.q0 = ss[0].rw;
  t.dq1 = ss[1].rw - ss[0].rw;
  t.dq2 = ss[2].rw - ss[0].rw;
  for (int k = 0; k < 8; k++) {
    float p0 = vv[0]->attr[k] * ss[0].rw;
    float p1 = vv[1]->attr[k] * ss[1].rw;
    float p2 = vv[2]->attr[k] * ss[2].rw;
    t.a0[k] = p0;
    t.da1[k] = p1 - p0;
    t.da2[k] = p2 - p0;
  }
  t.kind = cmd.kind;
  t.transparent = cmd.transparent;
  t.key = ps.cur_key;
  t.image = cmd.image;
  t.sampler = cmd.sampler;
  ps.tris.push_back(t);
  ps.binned++;
}

void process_triangle(PassState& ps, VOut v[3], const DrawCmd& cmd) {
  SvrContext* ctx = ps.ctx;
  float hw = (float)ctx->W * 0.5f, hh = (float)ctx->H * 0.5f;
  int c0 = outcode(v[0].clip), c1 = outcode(v[1].clip), c2 = outcode(v[2].clip);
  if (c0 & c1 & c2) return;
  if (((c0 | c1 | c2) & (OC_NEAR | OC_FAR)) == 0) {
    ScreenV s0 = to_screen(v[0].clip, hw, hh), s1 = to_screen(v[1].clip, hw, hh),
            s2 = to_screen(v[2].clip, hw, hh);
    if (s0.ok && s1.ok && s2.ok) {
      emit_triangle(ps, &v[0], &v[1], &v[2], s0, s1, s2, cmd);
      return;
    }
  }
  VOut poly[12];
  poly[0] = v[0];
  poly[1] = v[1];
  poly[2] = v[2];
  int n = clip_polygon(poly, 3);
  for (int i = 1; i + 1 < n; i++) {
    ScreenV s0 = to_screen(poly[0].clip, hw, hh), s1 = to_screen(poly[i].clip, hw, hh),
            s2 = to_screen(poly[i + 1].clip, hw, hh);
    if (!(s0.ok && s1.ok && s2.ok)) continue;
    if (!(poly[0].clip[3] > 0.0f && poly[i].clip[3] > 0.0f && poly[i + 1].clip[3] > 0.0f)) continue;
    emit_triangle(ps, &poly[0], &poly[i], &poly[i + 1], s0, s1, s2, cmd);
  }
}

// one draw: its triangles in index order, appended to ps (ps.cur_key = sequence number of the draw's first triangle - 1)
void run_draw(PassState& ps, const SvrSceneData* scene, const DrawCmd& cmd) {
  if (cmd.kind == PIPE_COLORED_TRIANGLE) {
    VOut v[3];
    for (int i = 0; i < 3; i++) colored_triangle_vert(i, v[i]);
    ps.cur_key++;
    process_triangle(ps, v, cmd);
    return;
  }
  float mvp[16];
  if (cmd.kind == PIPE_MESH) matmul4(scene->viewproj, cmd.mat, mvp);
  const Mesh& mesh = *cmd.mesh;
  uint32_t ntri = cmd.index_count / 3;
  for (uint32_t t = 0; t < ntri; t++) {
    VOut v[3];
    for (int k = 0; k < 3; k++) {
      const SvrVertex& vx = mesh.vtx[mesh.idx[cmd.first_index + 3 * t + k]];
      if (cmd.kind == PIPE_MESH)
        mesh_vert(vx, mvp, cmd.mat, cmd.material->color_factors, v[k]);
      else
        colored_triangle_mesh_vert(vx, cmd.mat, v[k]);
    }
    ps.cur_key++;
    process_triangle(ps, v, cmd);
  }
}

void run_geometry(PassState& ps, const SvrSceneData* scene, const std::vector<DrawCmd>& cmds) {
  for (const DrawCmd& cmd : cmds) run_draw(ps, scene, cmd);
}

// ---------------------------------------------------------------- fragment stage
inline void bary_at(const SetupTri& t, int px, int py, float& b1, float& b2) {
  int64_t e1 = t.A[1] * px + t.B[1] * py + t.Cu[1];
  int64_t e2 = t.A[2] * px + t.B[2] * py + t.Cu[2];
  b1 = (float)e1 * t.inv_area;
  b2 = (float)e2 * t.inv_area;
}
inline float interp(const SetupTri& t, int k, float b1, float b2, float r) {
  float ap = fmaf(b2, t.da2[k], fmaf(b1, t.da1[k], t.a0[k]));
  return ap * r;
}
inline float recip_w(const SetupTri& t, float b1, float b2) {
  float q = fmaf(b2, t.dq2, fmaf(b1, t.dq1, t.q0));
  return 1.0f / q;
}

// run the fragment shader of the triangle's pipeline at pixel (px,py); b1,b2 = its barycentrics
void shade(const SetupTri& t, const SvrSceneData* scene, int px, int py, float b1, float b2,
           float out[4], float* trace = nullptr) {
  float r = recip_w(t, b1, b2);
  if (t.kind == PIPE_COLORED_TRIANGLE) {  // shaders/colored_triangle.frag:9-12
    for (int c = 0; c < 3; c++) out[c] = interp(t, 3 + c, b1, b2, r);
    out[3] = 1.0f;
    return;
  }
  float u = interp(t, 6, b1, b2, r), v = interp(t, 7, b1, b2, r);
  // fine derivatives inside the 2x2 quad (C7): partners may lie outside the triangle (helper lanes)
  float hb1, hb2, vb1, vb2;
  bary_at(t, px ^ 1, py, hb1, hb2);
  bary_at(t, px, py ^ 1, vb1, vb2);
  float hr = recip_w(t, hb1, hb2), vr = recip_w(t, vb1, vb2);
  float uh = interp(t, 6, hb1, hb2, hr), vh = interp(t, 7, hb1, hb2, hr);
  float uv_ = interp(t, 6, vb1, vb2, vr), vv_ = interp(t, 7, vb1, vb2, vr);
  float dudx = (px & 1) ? (u - uh) : (uh - u);
  float dvdx = (px & 1) ? (v - vh) : (vh - v);
  float dudy = (py & 1) ? (u - uv_) : (uv_ - u);
  float dvdy = (py & 1) ? (v - vv_) : (vv_ - v);
  float tex[4];
  sample_texture(*t.image, *t.sampler, u, v, dudx, dvdx, dudy, dvdy, tex);
  if (trace) {  // slots shared with k_tile.hip's trace
    trace[0] = (float)t.key; trace[1] = b1; trace[2] = b2; trace[3] = r; trace[4] = u; trace[5] = v;
    trace[6] = dudx; trace[7] = dvdx; trace[8] = dudy; trace[9] = dvdy;
    for (int c = 0; c < 4; c++) trace[11 + c] = tex[c];
    trace[26] = hb1; trace[27] = hb2; trace[28] = vb1; trace[29] = vb2; trace[30] = hr; trace[31] = vr;
  }
  if (t.kind == PIPE_TEX_IMAGE) {  // shaders/tex_image.frag:10-12
    for (int c = 0; c < 4; c++) out[c] = tex[c];
    return;
  }
  // shaders/mesh.frag:12-19
  float nx = interp(t, 0, b1, b2, r), ny = interp(t, 1, b1, b2, r), nz = interp(t, 2, b1, b2, r);
  const float* L = scene->sunlight_direction;
  float d = fmaf(nz, L[2], fmaf(ny, L[1], nx * L[0]));
  float light = fmaxf(d, 0.1f);
  for (int c = 0; c < 3; c++) {
    float color = interp(t, 3 + c, b1, b2, r) * tex[c];
    float ambient = color * scene->ambient_color[c];
    out[c] = fmaf(color * light, scene->sunlight_color[3], ambient);
    if (trace) trace[18 + c] = color;
  }
  out[3] = 1.0f;
  if (trace) {
    trace[15] = nx; trace[16] = ny; trace[17] = nz; trace[21] = light;
    for (int c = 0; c < 4; c++) trace[22 + c] = out[c];
  }
}

inline void load_color(SvrContext* ctx, size_t p, float c[4]) {
  if (ctx->color_format == SVR_COLOR_RGBA16F)
    for (int k = 0; k < 4; k++) c[k] = f16_to_f32(ctx->color16[p * 4 + k]);
  else
    for (int k = 0; k < 4; k++) c[k] = unorm8_to_f32(ctx->color8[p * 4 + k]);
}
inline void store_color(SvrContext* ctx, size_t p, const float c[4]) {
  if (ctx->color_format == SVR_COLOR_RGBA16F)
    for (int k = 0; k < 4; k++) ctx->color16[p * 4 + k] = f32_to_f16(c[k]);
  else
    for (int k = 0; k < 4; k++) ctx->color8[p * 4 + k] = f32_to_unorm8(c[k]);
}

// svr_set_row_interleave: of the scissor's 32-row tile rows the context renders those with index % rstride == roff
inline bool owns_row(const SvrContext* ctx, uint32_t y) {
  return ctx->rstride == 1u || ((y - ctx->sy) >> 5) % ctx->rstride == ctx->roff;
}

// rasterise rows [y0,y1) of every triangle, in submission order
// subset (may be null): the triangles, in submission order, that reach rows y0 .. y1 - 1 — the threaded path hands
// every band its own list; submission order within a pixel is the order of `tris` either way.
void raster_rows(SvrContext* ctx, const SvrSceneData* scene, const std::vector<SetupTri>& tris,
                 int y0, int y1, uint64_t& n_raster, uint64_t& n_shaded, const std::vector<const SetupTri*>* subset = nullptr) {
  uint32_t W = ctx->W;
  const size_t count = subset ? subset->size() : tris.size();
  for (size_t ti = 0; ti < count; ti++) {
    const SetupTri& t = subset ? *(*subset)[ti] : tris[ti];
    int ya = std::max(t.miny, y0), yb = std::min(t.maxy, y1 - 1);
    for (int py = ya; py <= yb; py++) {
      if (!owns_row(ctx, (uint32_t)py)) continue;  // svr_set_row_interleave
      for (int px = t.minx; px <= t.maxx; px++) {
        int64_t e0 = t.A[0] * px + t.B[0] * py + t.C[0];
        int64_t e1 = t.A[1] * px + t.B[1] * py + t.C[1];
        int64_t e2 = t.A[2] * px + t.B[2] * py + t.C[2];
        if ((e0 | e1 | e2) < 0) continue;
        n_raster++;
        float b1, b2;
        bary_at(t, px, py, b1, b2);
        float z = fmaf(b2, t.dz2, fmaf(b1, t.dz1, t.z0));
        z = fminf(fmaxf(z, 0.0f), 1.0f) + 0.0f;  // + 0.0f: never store -0.0 (C5)
        size_t p = (size_t)py * W + px;
        // depth test GREATER_OR_EQUAL (src/vk_engine.cpp:1659)
        if (!(z >= ctx->depth[p])) continue;
        float src[4];
        bool tr = (px == ctx->trace_x && py == ctx->trace_y);
        shade(t, scene, px, py, b1, b2, src, tr ? ctx->trace : nullptr);
        n_shaded++;
        if (!t.transparent) {
          ctx->depth[p] = z;  // depth write on, blend off
          store_color(ctx, p, src);
        } else {
          // enable_blending_additive (src/vk_pipelines.cpp:157-167): rgb = src + dst*dst.a, a = src.a
          float dst[4], out[4];
          load_color(ctx, p, dst);
          for (int c = 0; c < 3; c++) out[c] = fmaf(dst[c], dst[3], src[c]);
          out[3] = src[3];
          if (tr)
            for (int c = 0; c < 4; c++) {
              ctx->trace[32 + c] = dst[c];
              ctx->trace[36 + c] = out[c];
            }
          store_color(ctx, p, out);
        }
      }
    }
  }
}

// fork-join over the oracle's worker count: fn(thread index) on n threads (n == 1: on the caller's)
template <class F>
void fork_join(int n, F fn) {
  if (n <= 1) {
    fn(0);
    return;
  }
  std::vector<std::thread> pool;
  for (int ti = 1; ti < n; ti++) pool.emplace_back([&fn, ti]() { fn(ti); });
  fn(0);
  for (auto& th : pool) th.join();
}
// rows [y0, y1) cut into n near-equal runs: run ti
inline void row_run(int y0, int y1, int n, int ti, int& a, int& b) {
  const int64_t rows = y1 - y0;
  a = y0 + (int)(rows * ti / n);
  b = y0 + (int)(rows * (ti + 1) / n);
}

int run_pass(SvrContext* ctx, const SvrSceneData* scene, const std::vector<DrawCmd>& cmds) {
  ctx->pass_sy = ctx->sy;
  ctx->pass_sh = ctx->sh;
  ctx->pass_rstride = ctx->rstride;
  PassState ps;
  ps.ctx = ctx;
  const int nthreads = std::max(1, ctx->threads);
  // Threaded (svr_oracle_set_threads: bench.py's all-core baseline): the draws are set up side by side, each into its own
  // list with the sequence numbers it has in the one-thread run; nothing is copied together afterwards — the bands of
  // rows below get lists of pointers in submission order (draw by draw), so every pixel still sees the one-thread order.
  std::vector<PassState> part;
  if (nthreads > 1 && cmds.size() > 1) {
    part.resize(cmds.size());
    uint32_t key = ps.cur_key;
    for (size_t i = 0; i < cmds.size(); i++) {
      part[i].ctx = ctx;
      part[i].cur_key = key;
      key += cmds[i].kind == PIPE_COLORED_TRIANGLE ? 1u : cmds[i].index_count / 3;
    }
    std::atomic<size_t> next_draw{0};
    fork_join(nthreads, [&](int) {
      for (;;) {
        size_t i = next_draw.fetch_add(1);
        if (i >= cmds.size()) break;
        run_draw(part[i], scene, cmds[i]);
      }
    });
    for (const PassState& q : part) ps.binned += q.binned;
  } else {
    run_geometry(ps, scene, cmds);
  }
  // depth loadOp CLEAR 0.0 over the render area = scissor here (src/vk_initializers.cpp:133-147)
  int y0 = (int)ctx->sy, y1 = (int)(ctx->sy + ctx->sh);
  fork_join(nthreads, [&](int ti) {
    int ya, yb;
    row_run(y0, y1, nthreads, ti, ya, yb);
    for (int y = ya; y < yb; y++)
      for (uint32_t x = ctx->sx; x < ctx->sx + ctx->sw; x++) ctx->depth[(size_t)y * ctx->W + x] = 0.0f;
  });
  uint64_t n_raster = 0, n_shaded = 0;
  if (nthreads == 1) {
    raster_rows(ctx, scene, ps.tris, y0, y1, n_raster, n_shaded);
  } else {
    const int band = 16;
    int nbands = (y1 - y0 + band - 1) / band;
    // Every band's triangles, in submission order.  The lists in submission order: ps.tris, then the draws' own.  They
    // are cut into nthreads runs of near-equal length; every thread counts what its run contributes to each band, a
    // prefix sum over the threads (per band) gives every run its place in every band's list, and the threads fill
    // them in: a band's list is run 0's entries, then run 1's, ... = submission order.  (One thread walking all lists
    // was a serial 3 % of the frame: enough to cap sixteen threads at 11x.)
    std::vector<const std::vector<SetupTri>*> lists;
    lists.push_back(&ps.tris);
    for (const PassState& q : part) lists.push_back(&q.tris);
    size_t n_all = 0;
    std::vector<size_t> list_base;
    for (auto* l : lists) {
      list_base.push_back(n_all);
      n_all += l->size();
    }
    auto for_run = [&](int ti, auto&& fn) {  // fn(triangle) over run ti of the concatenated lists
      size_t a = n_all * (size_t)ti / (size_t)nthreads, b = n_all * (size_t)(ti + 1) / (size_t)nthreads;
      size_t li = std::upper_bound(list_base.begin(), list_base.end(), a) - list_base.begin() - 1;
      for (size_t g = a; g < b;) {
        while (li + 1 < lists.size() && list_base[li + 1] <= g) li++;
        const std::vector<SetupTri>& l = *lists[li];
        size_t end = std::min(b, list_base[li] + l.size());
        for (; g < end; g++) fn(l[g - list_base[li]]);
      }
    };
    auto bands_of = [&](const SetupTri& t, int& b0, int& b1) {
      int ya = std::max(t.miny, y0), yb = std::min(t.maxy, y1 - 1);
      if (ya > yb || t.minx > t.maxx) return false;
      b0 = (ya - y0) / band;
      b1 = (yb - y0) / band;
      return true;
    };
    std::vector<std::vector<uint32_t>> cnt((size_t)nthreads, std::vector<uint32_t>((size_t)nbands, 0));
    fork_join(nthreads, [&](int ti) {
      std::vector<uint32_t>& c = cnt[(size_t)ti];
      for_run(ti, [&](const SetupTri& t) {
        int b0, b1;
        if (bands_of(t, b0, b1))
          for (int b = b0; b <= b1; b++) c[(size_t)b]++;
      });
    });
    std::vector<std::vector<const SetupTri*>> reach((size_t)nbands);
    for (int b = 0; b < nbands; b++) {
      uint32_t total = 0;
      for (int ti = 0; ti < nthreads; ti++) {
        uint32_t c = cnt[(size_t)ti][(size_t)b];
        cnt[(size_t)ti][(size_t)b] = total;  // now: where run ti starts in band b's list
        total += c;
      }
      reach[(size_t)b].resize(total);
    }
    fork_join(nthreads, [&](int ti) {
      std::vector<uint32_t>& at = cnt[(size_t)ti];
      for_run(ti, [&](const SetupTri& t) {
        int b0, b1;
        if (bands_of(t, b0, b1))
          for (int b = b0; b <= b1; b++) reach[(size_t)b][at[(size_t)b]++] = &t;
      });
    });
    std::atomic<int> next{0};
    std::vector<uint64_t> nr((size_t)nthreads, 0), ns((size_t)nthreads, 0);
    fork_join(nthreads, [&](int ti) {
      uint64_t my_raster = 0, my_shaded = 0;  // locals, stored once: the neighbours of nr[] / ns[] share a cache line, and
      for (;;) {                              // counting into them per fragment made two threads as slow as one
        int b = next.fetch_add(1);
        if (b >= nbands) break;
        int ya = y0 + b * band, yb = std::min(y1, ya + band);
        raster_rows(ctx, scene, ps.tris, ya, yb, my_raster, my_shaded, &reach[(size_t)b]);
      }
      nr[(size_t)ti] = my_raster;
      ns[(size_t)ti] = my_shaded;
    });
    for (int ti = 0; ti < nthreads; ti++) {
      n_raster += nr[(size_t)ti];
      n_shaded += ns[(size_t)ti];
    }
  }
  ctx->stats.rasterized_fragments = n_raster;
  ctx->stats.shaded_fragments = n_shaded;
  ctx->stats.binned_triangles = ps.binned;
  ctx->stats.bin_entries = 0;
  return SVR_OK;
}

// ---------------------------------------------------------------- is_visible, src/vk_engine.cpp:56-86
// glm operation order (glm is an un-vendored, unpinned submodule: the classic 0.9.9 scalar order
// is restated): mat4*mat4 column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3;
// mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3).  No fma (x86-64 -O3 without -mfma).
void glm_matmul(const float* a, const float* b, float* out) {
  for (int j = 0; j < 4; j++)
    for (int r = 0; r < 4; r++) {
      float acc = a[0 + r] * b[4 * j + 0];
      acc = acc + a[4 + r] * b[4 * j + 1];
      acc = acc + a[8 + r] * b[4 * j + 2];
      acc = acc + a[12 + r] * b[4 * j + 3];
      out[4 * j + r] = acc;
    }
}
void glm_matvec(const float* m, const float* v, float* out) {
  for (int r = 0; r < 4; r++) {
    float add0 = m[0 + r] * v[0] + m[4 + r] * v[1];
    float add1 = m[8 + r] * v[2] + m[12 + r] * v[3];
    out[r] = add0 + add1;
  }
}
bool is_visible(const SvrRenderObject& obj, const float* viewproj) {
  static const float corners[8][3] = {{1, 1, 1},  {1, 1, -1},  {1, -1, 1},  {1, -1, -1},
                                      {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
  float matrix[16];
  glm_matmul(viewproj, obj.transform, matrix);
  float mn[3] = {1.5f, 1.5f, 1.5f}, mx[3] = {-1.5f, -1.5f, -1.5f};
  for (int c = 0; c < 8; c++) {
    float p[4];
    for (int k = 0; k < 3; k++) p[k] = obj.bounds.origin[k] + corners[c][k] * obj.bounds.extents[k];
    p[3] = 1.0f;
    float v[4];
    glm_matvec(matrix, p, v);
    v[0] = v[0] / v[3];
    v[1] = v[1] / v[3];
    v[2] = v[2] / v[3];
    for (int k = 0; k < 3; k++) {
      mn[k] = std::min(v[k], mn[k]);  // glm::min(x, y) = (y < x) ? y : x
      mx[k] = std::max(v[k], mx[k]);
    }
  }
  if (mn[2] > 1.f || mx[2] < 0.f || mn[0] > 1.f || mx[0] < -1.f || mn[1] > 1.f || mx[1] < -1.f) return false;
  return true;
}

Mesh* get_mesh(SvrContext* ctx, SvrMesh h) {
  if (h == 0 || h > ctx->meshes.size() || !ctx->meshes[h - 1]->alive) return nullptr;
  return ctx->meshes[h - 1].get();
}
Image* get_image(SvrContext* ctx, SvrImage h) {
  if (h == 0 || h > ctx->images.size() || !ctx->images[h - 1]->alive) return nullptr;
  return ctx->images[h - 1].get();
}

}  // namespace

// ================================================================ C ABI
extern "C" {

const char* svr_last_error(void) { return g_err.c_str(); }
const char* svr_backend_name(void) { return "cpu-oracle"; }

int svr_create(const SvrConfig* cfg, SvrContext** out) {
  if (!cfg || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: null argument");
  if (cfg->width == 0 || cfg->height == 0 || cfg->width > 16384 || cfg->height > 16384)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: extent must be in 1..16384");
  if (cfg->color_format != SVR_COLOR_RGBA16F && cfg->color_format != SVR_COLOR_RGBA8)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create: unknown colour format");
  SvrContext* ctx = new SvrContext();
  ctx->W = cfg->width;
  ctx->H = cfg->height;
  ctx->color_format = cfg->color_format;
  size_t n = (size_t)ctx->W * ctx->H;
  if (ctx->color_format == SVR_COLOR_RGBA16F)
    ctx->color16.assign(n * 4, 0);
  else
    ctx->color8.assign(n * 4, 0);
  ctx->depth.assign(n, 0.0f);
  ctx->sx = ctx->sy = 0;
  ctx->sw = ctx->W;
  ctx->sh = ctx->H;
  *out = ctx;
  return SVR_OK;
}

void svr_destroy(SvrContext* ctx) { delete ctx; }

int svr_set_stream(SvrContext* ctx, void*) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  return SVR_OK;
}
int svr_bind_targets(SvrContext*, void*, void*) {
  return fail(SVR_ERR_UNSUPPORTED, "svr_bind_targets: the CPU oracle owns its targets");
}
int svr_get_targets(SvrContext*, void**, void**) {
  return fail(SVR_ERR_UNSUPPORTED, "svr_get_targets: the CPU oracle has no device targets");
}

int svr_upload_mesh(SvrContext* ctx, const uint32_t* indices, size_t n_indices,
                    const SvrVertex* vertices, size_t n_vertices, SvrMesh* out) {
  if (!ctx || !out || (!indices && n_indices) || (!vertices && n_vertices))
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_upload_mesh: null argument");
  for (size_t i = 0; i < n_indices; i++)
    if (indices[i] >= n_vertices) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_upload_mesh: index out of range");
  auto m = std::make_unique<Mesh>();
  m->idx.assign(indices, indices + n_indices);
  m->vtx.assign(vertices, vertices + n_vertices);
  m->alive = true;
  ctx->meshes.push_back(std::move(m));
  *out = (SvrMesh)ctx->meshes.size();
  return SVR_OK;
}
int svr_destroy_mesh(SvrContext* ctx, SvrMesh mesh) {
  Mesh* m = ctx ? get_mesh(ctx, mesh) : nullptr;
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_destroy_mesh: bad handle");
  m->alive = false;
  m->idx.clear();
  m->vtx.clear();
  return SVR_OK;
}

int svr_create_image(SvrContext* ctx, const void* rgba8, uint32_t width, uint32_t height,
                     int mipmapped, SvrImage* out) {
  if (!ctx || !rgba8 || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_image: null argument");
  if (width == 0 || height == 0 || width > 16384 || height > 16384)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_image: extent must be in 1..16384");
  auto img = std::make_unique<Image>();
  img->w = width;
  img->h = height;
  // mip count, src/vk_engine.cpp:1543-1545
  uint32_t levels = 1;
  if (mipmapped) {
    uint32_t m = std::max(width, height);
    while (m > 1) {
      m >>= 1;
      levels++;
    }
  }
  img->levels = levels;
  img->mip.resize(levels);
  img->lw.resize(levels);
  img->lh.resize(levels);
  img->lw[0] = width;
  img->lh[0] = height;
  img->mip[0].assign((const uint8_t*)rgba8, (const uint8_t*)rgba8 + (size_t)width * height * 4);
  for (uint32_t l = 1; l < levels; l++) {
    img->lw[l] = std::max(1u, img->lw[l - 1] >> 1);
    img->lh[l] = std::max(1u, img->lh[l - 1] >> 1);
    downsample_level(img->mip[l - 1], img->lw[l - 1], img->lh[l - 1], img->mip[l], img->lw[l], img->lh[l]);
  }
  img->alive = true;
  ctx->images.push_back(std::move(img));
  *out = (SvrImage)ctx->images.size();
  return SVR_OK;
}
int svr_destroy_image(SvrContext* ctx, SvrImage image) {
  Image* im = ctx ? get_image(ctx, image) : nullptr;
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_destroy_image: bad handle");
  im->alive = false;
  im->mip.clear();
  return SVR_OK;
}
int svr_read_image_level(SvrContext* ctx, SvrImage image, uint32_t level, void* dst, size_t bytes,
                         uint32_t* w, uint32_t* h) {
  Image* im = ctx ? get_image(ctx, image) : nullptr;
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_read_image_level: bad handle");
  if (level >= im->levels) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_image_level: no such level");
  if (w) *w = im->lw[level];
  if (h) *h = im->lh[level];
  if (dst) {
    if (bytes < im->mip[level].size()) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_image_level: buffer too small");
    std::memcpy(dst, im->mip[level].data(), im->mip[level].size());
  }
  return SVR_OK;
}

int svr_create_sampler(SvrContext* ctx, const SvrSamplerDesc* desc, SvrSampler* out) {
  if (!ctx || !desc || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: null argument");
  if ((desc->mag_filter | 1) != 1 || (desc->min_filter | 1) != 1 || (desc->mipmap_mode | 1) != 1)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: bad filter enum");
  if (!(desc->min_lod <= desc->max_lod)) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_create_sampler: min_lod > max_lod");
  Sampler s;
  s.d = *desc;
  ctx->samplers.push_back(s);
  *out = (SvrSampler)ctx->samplers.size();
  return SVR_OK;
}

int svr_write_material(SvrContext* ctx, int pass, const float color_factors[4],
                       const float metal_rough_factors[4], SvrImage color_image,
                       SvrSampler color_sampler, SvrMaterial* out) {
  if (!ctx || !color_factors || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_write_material: null argument");
  if (!get_image(ctx, color_image)) return fail(SVR_ERR_BAD_HANDLE, "svr_write_material: bad image");
  if (color_sampler == 0 || color_sampler > ctx->samplers.size()) return fail(SVR_ERR_BAD_HANDLE, "svr_write_material: bad sampler");
  if (pass < 0 || pass > 2) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_write_material: bad pass");
  Material m;
  m.pass = pass;
  for (int i = 0; i < 4; i++) {
    m.color_factors[i] = color_factors[i];
    m.metal_rough[i] = metal_rough_factors ? metal_rough_factors[i] : 0.0f;
  }
  m.image = color_image - 1;
  m.sampler = color_sampler - 1;
  ctx->materials.push_back(m);
  *out = (SvrMaterial)ctx->materials.size();
  return SVR_OK;
}

int svr_clear_color(SvrContext* ctx, const float rgba[4]) {
  if (!ctx || !rgba) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_clear_color: null argument");
  // whole rows of the scissor (= the whole target unless the multi-GPU path narrowed it)
  const int nthreads = std::max(1, ctx->threads);
  fork_join(nthreads, [&](int ti) {
    int ya, yb;
    row_run((int)ctx->sy, (int)(ctx->sy + ctx->sh), nthreads, ti, ya, yb);
    for (size_t p = (size_t)ya * ctx->W, p1 = (size_t)yb * ctx->W; p < p1; p++) store_color(ctx, p, rgba);
  });
  return SVR_OK;
}

// ---- draw_background (src/vk_engine.cpp:1341-1355): gradient_color.comp / sky.comp  (contract C14, C15)
namespace {

// cos for sky.comp's hash: three-term Cody-Waite reduction by pi/2 (fma), then the single-precision
// minimax polynomials for sin / cos on [-pi/4, pi/4].  Every operation is spelled out: the device
// kernel performs the same sequence, so the star field is bit-identical on both sides.
const float kTwoOverPi = 0x1.45f306p-1f;
const float kPio2Hi = 0x1.921fb6p+0f, kPio2Mid = -0x1.777a5cp-25f, kPio2Lo = -0x1.ee59dap-50f;
inline float sky_cos(float x) {
  float k = std::nearbyintf(x * kTwoOverPi);
  float r = std::fmaf(-k, kPio2Hi, x);
  r = std::fmaf(-k, kPio2Mid, r);
  r = std::fmaf(-k, kPio2Lo, r);
  float z = r * r;
  float sn = std::fmaf(std::fmaf(std::fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
  float cs = std::fmaf(std::fmaf(std::fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                       std::fmaf(-0.5f, z, 1.0f));
  int q = (int)k & 3;  // cos(r + q*pi/2)
  float v = (q & 1) ? sn : cs;
  return (q == 1 || q == 2) ? -v : v;
}
inline float sky_fract(float a) { return a - std::floor(a); }
inline float sky_noise2d(float x, float y) {  // shaders/sky.comp:17-22
  float xhash = sky_cos(x * 37.0f), yhash = sky_cos(y * 57.0f);
  return sky_fract(415.92653f * (xhash + yhash));
}
inline float sky_noisy_star(float x, float y, float thr) {  // shaders/sky.comp:25-33
  float s = sky_noise2d(x, y);
  if (!(s >= thr)) return 0.0f;
  float t = (s - thr) / (1.0f - thr);
  float t2 = t * t, t4 = t2 * t2;  // pow(t, 6.0)
  return t4 * t2;
}
inline float sky_stable_star(float x, float y, float thr) {  // shaders/sky.comp:36-54
  float fx = sky_fract(x), fy = sky_fract(y);
  float gx = std::floor(x), gy = std::floor(y);
  float v1 = sky_noisy_star(gx, gy, thr), v2 = sky_noisy_star(gx, gy + 1.0f, thr);
  float v3 = sky_noisy_star(gx + 1.0f, gy, thr), v4 = sky_noisy_star(gx + 1.0f, gy + 1.0f, thr);
  float acc = (v1 * (1.0f - fx)) * (1.0f - fy);
  acc = std::fmaf(v2 * (1.0f - fx), fy, acc);
  acc = std::fmaf(v3 * fx, 1.0f - fy, acc);
  acc = std::fmaf(v4 * fx, fy, acc);
  return acc;
}

}  // namespace

int svr_draw_background(SvrContext* ctx, int effect, const float data[16]) {
  if (!ctx || !data) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_background: null argument");
  if (effect != SVR_BACKGROUND_GRADIENT && effect != SVR_BACKGROUND_SKY)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_background: unknown effect");
  const float fh = (float)ctx->H;
  for (uint32_t y = ctx->sy; y < ctx->sy + ctx->sh; y++)
    for (uint32_t x = 0; x < ctx->W; x++) {
      float out[4];
      if (effect == SVR_BACKGROUND_GRADIENT) {  // shaders/gradient_color.comp:19-26
        float blend = (float)y / fh;
        for (int c = 0; c < 4; c++) out[c] = std::fmaf(data[4 + c], blend, data[c] * (1.0f - blend));
      } else {  // shaders/sky.comp:56-76 (fragCoord = the integer texel coordinate)
        float fx = (float)x, fy = (float)y;
        float star = sky_stable_star(fx + 0.2f, fy + -0.06f, data[3]);
        for (int c = 0; c < 3; c++) out[c] = (data[c] * fy) / fh + star;
        out[3] = 1.0f;
      }
      store_color(ctx, (size_t)y * ctx->W + x, out);
    }
  return SVR_OK;
}

// ---- vkutil::copy_image (src/vk_images.cpp:33-64): LINEAR blit to the swapchain format  (contract C16)
static int blit_to(SvrContext* ctx, uint32_t dw, uint32_t dh, int fmt, uint8_t* dst, uint32_t row_first, uint32_t n_rows,
                   bool own_rows_only = false) {
  const float su = (float)ctx->W / (float)dw, sv = (float)ctx->H / (float)dh;
  for (uint32_t j = row_first; j < row_first + n_rows; j++) {
    if (own_rows_only && !owns_row(ctx, j)) continue;
    for (uint32_t i = 0; i < dw; i++) {
      float u = ((float)i + 0.5f) * su - 0.5f, v = ((float)j + 0.5f) * sv - 0.5f;
      float fu = std::floor(u), fv = std::floor(v);
      float a = u - fu, b = v - fv;
      long i0 = (long)fu, j0 = (long)fv, i1 = i0 + 1, j1 = j0 + 1;
      auto clampi = [](long t, long hi) { return t < 0 ? 0 : (t > hi ? hi : t); };
      i0 = clampi(i0, (long)ctx->W - 1); i1 = clampi(i1, (long)ctx->W - 1);
      j0 = clampi(j0, (long)ctx->H - 1); j1 = clampi(j1, (long)ctx->H - 1);
      float t00[4], t10[4], t01[4], t11[4], o[4];
      load_color(ctx, (size_t)j0 * ctx->W + i0, t00);
      load_color(ctx, (size_t)j0 * ctx->W + i1, t10);
      load_color(ctx, (size_t)j1 * ctx->W + i0, t01);
      load_color(ctx, (size_t)j1 * ctx->W + i1, t11);
      for (int c = 0; c < 4; c++) {
        float top = std::fmaf(a, t10[c] - t00[c], t00[c]), bot = std::fmaf(a, t11[c] - t01[c], t01[c]);
        o[c] = std::fmaf(b, bot - top, top);
      }
      uint8_t* d = dst + ((size_t)j * dw + i) * 4;
      if (fmt == SVR_SWAPCHAIN_B8G8R8A8) {
        d[0] = f32_to_unorm8(o[2]); d[1] = f32_to_unorm8(o[1]); d[2] = f32_to_unorm8(o[0]); d[3] = f32_to_unorm8(o[3]);
      } else {
        for (int c = 0; c < 4; c++) d[c] = f32_to_unorm8(o[c]);
      }
    }
  }
  return SVR_OK;
}

int svr_read_swapchain(SvrContext* ctx, uint32_t dw, uint32_t dh, int fmt, void* dst, size_t bytes) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_swapchain: null argument");
  if (dw == 0 || dh == 0 || dw > 16384 || dh > 16384) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_swapchain: extent must be in 1..16384");
  if (fmt != SVR_SWAPCHAIN_B8G8R8A8 && fmt != SVR_SWAPCHAIN_R8G8B8A8)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_swapchain: unknown format");
  if (bytes < (size_t)dw * dh * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_swapchain: buffer too small");
  return blit_to(ctx, dw, dh, fmt, (uint8_t*)dst, 0, dh);
}

// the oracle has no device memory: the "swapchain image" is host memory here
int svr_copy_to_swapchain(SvrContext* ctx, void* dst, uint32_t dw, uint32_t dh, int fmt) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_copy_to_swapchain: null argument");
  if (dw == 0 || dh == 0 || dw > 16384 || dh > 16384) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_copy_to_swapchain: extent must be in 1..16384");
  if (fmt != SVR_SWAPCHAIN_B8G8R8A8 && fmt != SVR_SWAPCHAIN_R8G8B8A8)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_copy_to_swapchain: unknown format");
  const bool identity = dw == ctx->W && dh == ctx->H;  // identity extent: the scissor's rows only
  if (ctx->present_status) *ctx->present_status = 0u;  // nothing here is ever void
  return blit_to(ctx, dw, dh, fmt, (uint8_t*)dst, identity ? ctx->sy : 0u, identity ? ctx->sh : dh, identity);
}

int svr_set_row_interleave(SvrContext* ctx, uint32_t stride, uint32_t offset) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (stride == 0 || stride > 64 || offset >= stride) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_set_row_interleave: need 1 <= stride <= 64, offset < stride");
  ctx->rstride = stride;
  ctx->roff = offset;
  return SVR_OK;
}

int svr_set_present_status(SvrContext* ctx, uint32_t* status) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  ctx->present_status = status;
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

static int validate_object(SvrContext* ctx, const SvrRenderObject& o, const char* which) {
  Mesh* m = get_mesh(ctx, o.mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, std::string("svr_draw_geometry: bad mesh handle in ") + which);
  if (o.material == 0 || o.material > ctx->materials.size())
    return fail(SVR_ERR_BAD_HANDLE, std::string("svr_draw_geometry: bad material handle in ") + which);
  if ((uint64_t)o.first_index + o.index_count > m->idx.size())
    return fail(SVR_ERR_INVALID_ARGUMENT, std::string("svr_draw_geometry: index range outside the mesh in ") + which);
  return SVR_OK;
}

int svr_draw_geometry(SvrContext* ctx, const SvrSceneData* scene, const SvrRenderObject* opaque,
                      size_t n_opaque, const SvrRenderObject* transparent, size_t n_transparent,
                      SvrStats* out_stats) {
  if (!ctx || !scene || (!opaque && n_opaque) || (!transparent && n_transparent))
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: null argument");
  auto t0 = std::chrono::steady_clock::now();
  // MeshNode::Draw routes by pass_type (src/vk_engine.cpp:1729-1733): Transparent materials only in
  // the transparent list, everything else only in the opaque list.  Anything else is rejected.
  for (size_t i = 0; i < n_opaque; i++) {
    if (int e = validate_object(ctx, opaque[i], "opaque")) return e;
    if (ctx->materials[opaque[i].material - 1].pass == SVR_PASS_TRANSPARENT)
      return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: Transparent material in the opaque list");
  }
  for (size_t i = 0; i < n_transparent; i++) {
    if (int e = validate_object(ctx, transparent[i], "transparent")) return e;
    if (ctx->materials[transparent[i].material - 1].pass != SVR_PASS_TRANSPARENT)
      return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_geometry: non-Transparent material in the transparent list");
  }
  // cull (src/vk_engine.cpp:1361-1367): opaque only
  std::vector<uint32_t> order;
  order.reserve(n_opaque);
  for (size_t i = 0; i < n_opaque; i++)
    if (is_visible(opaque[i], scene->viewproj)) order.push_back((uint32_t)i);
  // sort (src/vk_engine.cpp:1369-1378).  The reference compares pointers/handles with an unstable
  // sort; the deterministic restatement is (material handle, mesh handle, submission index).
  std::stable_sort(order.begin(), order.end(), [&](uint32_t ia, uint32_t ib) {
    const SvrRenderObject& a = opaque[ia];
    const SvrRenderObject& b = opaque[ib];
    if (a.material == b.material) return a.mesh < b.mesh;
    return a.material < b.material;
  });
  std::vector<DrawCmd> cmds;
  cmds.reserve(order.size() + n_transparent);
  auto push = [&](const SvrRenderObject& o) {
    DrawCmd c{};
    c.kind = PIPE_MESH;
    c.mesh = get_mesh(ctx, o.mesh);
    c.first_index = o.first_index;
    c.index_count = o.index_count;
    std::memcpy(c.mat, o.transform, sizeof(c.mat));
    c.material = &ctx->materials[o.material - 1];
    c.image = ctx->images[c.material->image].get();
    c.sampler = &ctx->samplers[c.material->sampler];
    // pipeline comes from the material (src/vk_engine.cpp:1695-1699), not from the list it is in
    c.transparent = (c.material->pass == SVR_PASS_TRANSPARENT);
    cmds.push_back(c);
  };
  SvrStats st{};
  for (uint32_t i : order) {
    push(opaque[i]);
    st.drawcall_count++;
    st.triangle_count += (int)(opaque[i].index_count / 3);
  }
  for (size_t i = 0; i < n_transparent; i++) {
    push(transparent[i]);
    st.drawcall_count++;
    st.triangle_count += (int)(transparent[i].index_count / 3);
  }
  st.culled_draws = (uint32_t)(n_opaque - order.size());
  ctx->stats = st;
  int e = run_pass(ctx, scene, cmds);
  auto t1 = std::chrono::steady_clock::now();
  ctx->stats.mesh_draw_time = std::chrono::duration<float, std::milli>(t1 - t0).count();
  ctx->stats.gpu_time_ms = ctx->stats.mesh_draw_time;
  if (out_stats) *out_stats = ctx->stats;
  return e;
}

int svr_draw_colored_triangle(SvrContext* ctx, SvrStats* out_stats) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  std::vector<DrawCmd> cmds(1);
  cmds[0] = DrawCmd{};
  cmds[0].kind = PIPE_COLORED_TRIANGLE;
  cmds[0].index_count = 3;
  SvrStats st{};
  st.drawcall_count = 1;
  st.triangle_count = 1;
  ctx->stats = st;
  SvrSceneData scene{};
  int e = run_pass(ctx, &scene, cmds);
  if (out_stats) *out_stats = ctx->stats;
  return e;
}

int svr_draw_tex_image(SvrContext* ctx, SvrMesh mesh, uint32_t first_index, uint32_t index_count,
                       const float render_matrix[16], SvrImage image, SvrSampler sampler,
                       SvrStats* out_stats) {
  if (!ctx || !render_matrix) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_tex_image: null argument");
  Mesh* m = get_mesh(ctx, mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad mesh");
  Image* im = get_image(ctx, image);
  if (!im) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad image");
  if (sampler == 0 || sampler > ctx->samplers.size()) return fail(SVR_ERR_BAD_HANDLE, "svr_draw_tex_image: bad sampler");
  if ((uint64_t)first_index + index_count > m->idx.size())
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_draw_tex_image: index range outside the mesh");
  std::vector<DrawCmd> cmds(1);
  cmds[0] = DrawCmd{};
  cmds[0].kind = PIPE_TEX_IMAGE;
  cmds[0].mesh = m;
  cmds[0].first_index = first_index;
  cmds[0].index_count = index_count;
  std::memcpy(cmds[0].mat, render_matrix, sizeof(float) * 16);
  cmds[0].image = im;
  cmds[0].sampler = &ctx->samplers[sampler - 1];
  SvrStats st{};
  st.drawcall_count = 1;
  st.triangle_count = (int)(index_count / 3);
  ctx->stats = st;
  SvrSceneData scene{};
  int e = run_pass(ctx, &scene, cmds);
  if (out_stats) *out_stats = ctx->stats;
  return e;
}

int svr_run_mesh_vert(SvrContext* ctx, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices,
                      const float world[16], const SvrSceneData* scene, SvrMaterial material,
                      float* out_clip, float* out_varyings) {
  if (!ctx || !world || !scene || !out_clip || !out_varyings)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_mesh_vert: null argument");
  Mesh* m = get_mesh(ctx, mesh);
  if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_run_mesh_vert: bad mesh");
  if (material == 0 || material > ctx->materials.size()) return fail(SVR_ERR_BAD_HANDLE, "svr_run_mesh_vert: bad material");
  if ((uint64_t)first_vertex + n_vertices > m->vtx.size())
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_mesh_vert: vertex range outside the mesh");
  float mvp[16];
  matmul4(scene->viewproj, world, mvp);
  const Material& mat = ctx->materials[material - 1];
  for (uint32_t i = 0; i < n_vertices; i++) {
    VOut o;
    mesh_vert(m->vtx[first_vertex + i], mvp, world, mat.color_factors, o);
    std::memcpy(out_clip + 4 * (size_t)i, o.clip, 16);
    std::memcpy(out_varyings + 8 * (size_t)i, o.attr, 32);
  }
  return SVR_OK;
}

int svr_run_vertex_shader(SvrContext* ctx, int shader, SvrMesh mesh, uint32_t first_vertex, uint32_t n_vertices,
                          const float render_matrix[16], float* out_clip, float* out_varyings) {
  if (!ctx || !out_clip || !out_varyings) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: null argument");
  const Mesh* m = nullptr;
  if (shader == SVR_VS_COLORED_TRIANGLE) {
    if ((uint64_t)first_vertex + n_vertices > 3) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: colored_triangle.vert has 3 vertices");
  } else if (shader == SVR_VS_COLORED_TRIANGLE_MESH) {
    if (!render_matrix) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: null matrix");
    m = get_mesh(ctx, mesh);
    if (!m) return fail(SVR_ERR_BAD_HANDLE, "svr_run_vertex_shader: bad mesh");
    if ((uint64_t)first_vertex + n_vertices > m->vtx.size())
      return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: vertex range outside the mesh");
  } else {
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_run_vertex_shader: unknown shader");
  }
  for (uint32_t i = 0; i < n_vertices; i++) {
    VOut o;
    if (m) colored_triangle_mesh_vert(m->vtx[first_vertex + i], render_matrix, o);
    else colored_triangle_vert((int)(first_vertex + i), o);
    std::memcpy(out_clip + 4 * (size_t)i, o.clip, 16);
    std::memcpy(out_varyings + 8 * (size_t)i, o.attr, 32);
  }
  return SVR_OK;
}

int svr_set_option(SvrContext* ctx, int option, int64_t) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  if (option != SVR_OPT_COUNT_FRAGMENTS && option != SVR_OPT_KERNEL_TIMING && option != SVR_OPT_TILE_CYCLES &&
      option != SVR_OPT_TUNING && option != SVR_OPT_QUEUE_CAPS && option != SVR_OPT_DEVICE_FLATTEN)
    return fail(SVR_ERR_INVALID_ARGUMENT, "svr_set_option: unknown option");
  return SVR_OK;  // the oracle always counts and has no kernels to time
}

int svr_debug_trace_pixel(SvrContext* ctx, int x, int y) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  ctx->trace_x = x;
  ctx->trace_y = y;
  std::memset(ctx->trace, 0, sizeof(ctx->trace));
  return SVR_OK;
}
int svr_debug_read_trace(SvrContext* ctx, float out[64]) {
  if (!ctx || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_debug_read_trace: null argument");
  std::memcpy(out, ctx->trace, sizeof(ctx->trace));
  return SVR_OK;
}

int svr_debug_read_bins(SvrContext*, uint32_t*, size_t, uint32_t*) {
  return fail(SVR_ERR_UNSUPPORTED, "svr_debug_read_bins: the CPU oracle does not bin");
}

int svr_get_row_costs(SvrContext* ctx, uint32_t* costs, size_t capacity, uint32_t* n_tile_rows, uint32_t* first_row, uint32_t* n_rows) {
  if (!ctx || !n_tile_rows) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_row_costs: null argument");
  uint32_t n = (ctx->pass_sh + 31u) / 32u;  // the oracle has no bins to weigh: every tile row costs the same
  n = n > ctx->roff ? (n - ctx->roff + ctx->pass_rstride - 1u) / ctx->pass_rstride : 0u;  // the context's own tile rows
  *n_tile_rows = n;
  if (first_row) *first_row = ctx->pass_sy;
  if (n_rows) *n_rows = ctx->pass_sh;
  if (costs) {
    if (capacity < n) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_row_costs: buffer too small");
    for (uint32_t i = 0; i < n; i++) costs[i] = 1u;
  }
  return SVR_OK;
}

int svr_debug_rcp_sweep(SvrContext*, int, uint64_t, uint64_t, uint64_t*, uint64_t*, uint32_t*) {
  return fail(SVR_ERR_UNSUPPORTED, "svr_debug_rcp_sweep: the CPU oracle divides (1.0f / x) everywhere");
}

int svr_debug_read_tile_cycles(SvrContext*, uint32_t*, size_t) {
  return fail(SVR_ERR_UNSUPPORTED, "svr_debug_read_tile_cycles: the CPU oracle has no tiles");
}

int svr_sync(SvrContext* ctx) {
  if (!ctx) return fail(SVR_ERR_INVALID_ARGUMENT, "null context");
  return SVR_OK;
}

int svr_read_color(SvrContext* ctx, void* dst, size_t bytes, int as_rgba8) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: null argument");
  size_t n = (size_t)ctx->W * ctx->H;
  if (ctx->color_format == SVR_COLOR_RGBA8) {
    if (bytes < n * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: buffer too small");
    std::memcpy(dst, ctx->color8.data(), n * 4);
    return SVR_OK;
  }
  if (!as_rgba8) {
    if (bytes < n * 8) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: buffer too small");
    std::memcpy(dst, ctx->color16.data(), n * 8);
    return SVR_OK;
  }
  if (bytes < n * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_color: buffer too small");
  uint8_t* d = (uint8_t*)dst;
  for (size_t i = 0; i < n * 4; i++) d[i] = f32_to_unorm8(f16_to_f32(ctx->color16[i]));
  return SVR_OK;
}

int svr_read_depth(SvrContext* ctx, float* dst, size_t bytes) {
  if (!ctx || !dst) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_depth: null argument");
  size_t n = (size_t)ctx->W * ctx->H;
  if (bytes < n * 4) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_read_depth: buffer too small");
  std::memcpy(dst, ctx->depth.data(), n * 4);
  return SVR_OK;
}

int svr_get_stats(SvrContext* ctx, SvrStats* out) {
  if (!ctx || !out) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_get_stats: null argument");
  *out = ctx->stats;
  return SVR_OK;
}

// ---- oracle-only extras (not part of svr.h)
int svr_oracle_set_threads(SvrContext* ctx, int n) {
  if (!ctx || n < 1) return fail(SVR_ERR_INVALID_ARGUMENT, "svr_oracle_set_threads: bad argument");
  ctx->threads = n;
  return SVR_OK;
}
int svr_oracle_is_visible(const SvrRenderObject* obj, const float viewproj[16]) {
  return is_visible(*obj, viewproj) ? 1 : 0;
}
uint16_t svr_oracle_f32_to_f16(float f) { return f32_to_f16(f); }
float svr_oracle_f16_to_f32(uint16_t h) { return f16_to_f32(h); }
float svr_oracle_lod(float rho2) { return lod_from_rho2(rho2); }

}  // extern "C"
