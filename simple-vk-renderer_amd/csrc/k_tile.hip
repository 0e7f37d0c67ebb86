// k_tile.hip — per-tile rasterisation, depth test, fragment shading, blending and write-back.
//
// Replaces everything the Vulkan implementation does per fragment for the two material pipelines
// (src/vk_engine.cpp:1619-1688): fill rasterisation at pixel centres with the top-left rule, depth
// test GREATER_OR_EQUAL on D32 with write (opaque) or without (transparent), mesh.frag
// (shaders/mesh.frag:12-19) / tex_image.frag / colored_triangle.frag with implicit-LOD texturing,
// blending (src/vk_pipelines.cpp:151-167), the depth clear to 0.0 and colour LOAD/STORE
// (src/vk_initializers.cpp:117-164).
//
// One 256-thread workgroup per 32x32 tile, launched heaviest tile first.  Pixel ownership: wave w owns
// the 16x16 quadrant w and lane l four pixels of it (one per 8x8 block); from the end of phase A a lane holds its four
// winning record indices, the tile's depth and colour live in LDS.
//   phase A  opaque visibility by column scan: the bin is staged through LDS 64 records at a time, every
//            triangle expands into one work item per pixel column of its bbox, a lane walks its column
//            with exact fp64 edge increments and resolves visibility with ds_max_u64 on an LDS
//            (depth bits << 32 | key) tile: per pixel max over (depth, submission key) == in-order
//            GREATER_OR_EQUAL with depth write, so bins need no order
//   phase B  shade each visible pixel once (deferred: identical result, no overdraw shading); a
//            specialised instance serves waves whose pixels all carry the key's common-case bit
//   phase C  transparent bin sorted by key in LDS, banded column scan (wave w owns rows 8w..8w+7) that
//            appends depth-passing fragments to per-wave queues in submission order; the scan stops whenever a
//            queue holds 64 fragments, they are shaded and blended at target precision, and the scan resumes from a
//            (chunk, row) pair; bins over SORT_CAP (1392) entries sort in a global arena instead
//   phase D  write-back: whole tiles go out through LDS as full rows, the depth CLEAR and a deferred
//            svr_clear_color are fused here
// Heavy tiles of small passes are rendered as four 8-row quarters by four workgroups (tile_kernel<.., SPLIT>).
// What bounds it is memory latency and VALU issue (the fragment stage is ~350 instructions per pixel), not HBM:
// DESIGN.md "Tile kernel".  <= 96 VGPRs and < 32 KiB of LDS = 5 workgroups per CU (round 3; 124 / 37 KiB = 4 before).
#include <hip/hip_fp16.h>

#include <hip/hip_ext.h>

#include <cstdlib>

#include "svr_launch.h"

namespace svr {

constexpr uint32_t REC_COMMON = 0x80000000u;  // in a lane's winning record index: the key's common-case bit (record indices stay below 2^31)
#ifndef SVR_PRIO_COST
#define SVR_PRIO_COST 200u  // tile_cost (thousands of cycles, svr_device.h) from which a tile's waves run at raised priority
#endif
constexpr int BATCH = 64;  // triangles staged per LDS batch == wave size
// A staged record is the SEVEN 16-byte pieces the scans read (box/key/flags, depth plane, nine edge coefficients), 112
// bytes apart: at the records' own 128 bytes every lane that reads the same piece of a different record falls on the
// same four LDS banks (a sliver-rich batch: twenty-way conflicts on each of the seven reads per work item; the tile
// kernel spent 8.0 M of its 18.6 M LDS cycles per launch on bank conflicts); at 112 eight records tile the 32 banks.
constexpr uint32_t SREC = 7;  // pieces (uint4) per staged record

// ------------------------------------------------------------------------------------------------
// texture unit (C8, C9)
constexpr float kInv255 = 0x1.010102p-8f;
constexpr float kG0 = 0x1.c51282p-2f, kG1 = -0x1.14a3a2p-2f, kG2 = 0x1.37536ap-3f, kG3 = -0x1.778474p-5f;

__device__ __forceinline__ float lod_from_rho2(float rho2) {
  uint32_t bits = f2u(rho2);
  int e = (int)(bits >> 23) - 127;
  float m = u2f((bits & 0x7fffffu) | 0x3f800000u);
  float t = m - 1.0f;
  float g = fmaf(fmaf(fmaf(kG3, t, kG2), t, kG1), t, kG0);
  float s = t * (1.0f - t);
  float l = fmaf(s, g, t);
  float lam = 0.5f * ((float)e + l);
  lam = (rho2 <= 0x1p+100f) ? lam : 50.0f;
  lam = (rho2 >= 0x1p-100f) ? lam : -50.0f;  // also NaN
  return lam;
}

__device__ __forceinline__ float lerpf(float a, float b, float t) { return fmaf(t, b - a, a); }

struct TexD {
  uint32_t base_off;  // level 0 in the texel arena
  uint32_t w, h, lw, lh, levels, filters;
  float min_lod, max_lod;
};

// the four texel byte offsets (from the texel arena's base) and the two weights of one mip level; NEAREST is
// the same footprint with both taps on floor(U) and weight 0 (lerp(t,t,0) == t exactly), so the path is branch-free
struct Taps {
  uint32_t o00, o10, o01, o11;
  float alpha, beta;
};
// wl, hl: the level's extent; base: its byte offset in the arena.
// POW2: the image's extents are powers of two (part of the COMMON case): REPEAT is a mask
template <bool POW2 = false>
__device__ __forceinline__ Taps level_taps(int wl, int hl, uint32_t base, bool linear, float u, float v) {
  float U = (u - floorf(u)) * (float)wl;
  float V = (v - floorf(v)) * (float)hl;
  float Uh = U - 0.5f, Vh = V - 0.5f;
  float fu = floorf(linear ? Uh : U), fv = floorf(linear ? Vh : V);
  Taps tp;
  tp.alpha = linear ? Uh - fu : 0.0f;
  tp.beta = linear ? Vh - fv : 0.0f;
  int i0 = (int)fu, j0 = (int)fv;
  int i1 = linear ? i0 + 1 : i0, j1 = linear ? j0 + 1 : j0;
  if (POW2) {  // indices are in [-1, extent]: one wrap either way == the mask
    i0 &= wl - 1;
    i1 &= wl - 1;
    j0 &= hl - 1;
    j1 &= hl - 1;
  } else {
    if (i0 < 0) i0 += wl;
    if (i0 >= wl) i0 -= wl;
    if (i1 >= wl) i1 -= wl;
    if (j0 < 0) j0 += hl;
    if (j0 >= hl) j0 -= hl;
    if (j1 >= hl) j1 -= hl;
  }
  // rows and extents are below 2^14: the 24-bit multiply is full rate where v_mul_lo_u32 is quarter rate
  const uint32_t r0 = __umul24((uint32_t)j0, (uint32_t)wl), r1 = __umul24((uint32_t)j1, (uint32_t)wl);
  tp.o00 = base + (r0 + (uint32_t)i0) * 4u;
  tp.o10 = base + (r0 + (uint32_t)i1) * 4u;
  tp.o01 = base + (r1 + (uint32_t)i0) * 4u;
  tp.o11 = base + (r1 + (uint32_t)i1) * 4u;
  return tp;
}
// a texel: the arena's base is wave-uniform (a kernel argument), the offset 32 bits per lane — a global load
// with an SGPR base, no 64-bit address arithmetic
__device__ __forceinline__ uint32_t texel(const uint8_t* arena, uint32_t off) {
  return *reinterpret_cast<const uint32_t*>(arena + off);
}
__device__ __forceinline__ float chan(uint32_t t, int c) { return (float)((t >> (8 * c)) & 0xffu) * kInv255; }
__device__ __forceinline__ float bilerp(uint32_t t00, uint32_t t10, uint32_t t01, uint32_t t11, int c, float a, float b) {
  return lerpf(lerpf(chan(t00, c), chan(t10, c), a), lerpf(chan(t01, c), chan(t11, c), a), b);
}

// ------------------------------------------------------------------------------------------------
// fragment stage: run the triangle's fragment shader at pixel (px,py)  (C6, C7, C10, C11)
__device__ __forceinline__ float interp3(float a0, float da1, float da2, float b1, float b2, float r) {
  return fmaf(b2, da2, fmaf(b1, da1, a0)) * r;
}

// The quad partners' u,v by DPP: lanes ^1 / ^8 of an 8x8 block hold the pixel's horizontal / vertical quad partner.
// Where every quad of the block is of one triangle (QUADS, wave-uniform: 42 % of a 4K frame's pixels, 77 % of the 8K x16
// frame's) the partner's own u,v ARE the values this lane would extrapolate for it — the edge functions are exact
// integers either way, the rest is the same chain on the same record — and two barycentric pairs, two reciprocals
// and four attribute interpolations per pixel are not computed.
__device__ __forceinline__ float quad_partner_x(float v) { return u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v), 0xB1, 0xf, 0xf, true)); }   // quad_perm:[1,0,3,2]
__device__ __forceinline__ float quad_partner_y(float v) { return u2f((uint32_t)__builtin_amdgcn_mov_dpp((int)f2u(v), 0x128, 0xf, 0xf, true)); }  // row_ror:8
// COMMON: the caller knows (key bit 1) that this is mesh.frag with a LINEAR/LINEAR/MIPMAP_LINEAR sampler
// on an image with power-of-two extents:
// pipeline kind, filter and mip-mode selections fold away (same arithmetic on the surviving path).
template <bool TRACE, bool COMMON = false>
__device__ __forceinline__ float4 shade_pixel(const FrameParams& P, uint32_t rec, int px, int py, float* trace, const bool quads = false) {
  const TriRec* tr = P.recs + rec;
  const uint4* q4 = reinterpret_cast<const uint4*>(tr);
  const float4* f4 = reinterpret_cast<const float4*>(tr);
  const double2* d2 = reinterpret_cast<const double2*>(tr);
  // every load of the record is issued up front: nothing below depends on another load except the texels
  uint4 hdr = q4[0];
  float4 zrow = f4[1];
  double2 c2 = d2[2], c3 = d2[3], c4 = d2[4], c5 = d2[5], c6 = d2[6];
  uint4 td = q4[7];
  float4 s0 = f4[8], s1 = f4[9], s2 = f4[10], s3 = f4[11], s4 = f4[12], s5 = f4[13], s6 = f4[14];
  // s0 = q0,dq1,dq2,a0[0] | s1 = a0[1..4] | s2 = a0[5..7],da1[0] | s3 = da1[1..4] | s4 = da1[5..7],da2[0]
  // s5 = da2[1..4] | s6 = da2[5..7],pad
  uint32_t flags = hdr.w;
  float inv_area = zrow.w;
  double A1, B1, A2, B2, e1, e2;
  double C1 = c5.y, C2 = c6.x;
  A1 = c2.y; B1 = c4.x; A2 = c3.x; B2 = c4.y;
  double dx = (double)px, dy = (double)py;
  e1 = fma(A1, dx, fma(B1, dy, C1)) + ((flags & F_T1) ? 1.0 : 0.0);
  e2 = fma(A2, dx, fma(B2, dy, C2)) + ((flags & F_T2) ? 1.0 : 0.0);
  TexD t;
  t.base_off = (uint32_t)(unsigned long long)__double_as_longlong(c6.y);
  t.w = td.x & 0xffffu;
  t.h = td.x >> 16;
  t.lw = td.y & 0xffu;
  t.lh = (td.y >> 8) & 0xffu;
  t.levels = (td.y >> 16) & 0xffu;
  t.filters = td.y >> 24;
  t.min_lod = u2f(td.z);
  t.max_lod = u2f(td.w);
  // e1,e2: unbiased edge values at the pixel; its horizontal / vertical quad partners are +-A, +-B away
  float b1 = (float)e1 * inv_area, b2 = (float)e2 * inv_area;
  float q0 = s0.x, dq1 = s0.y, dq2 = s0.z;
  const float qq = fmaf(b2, dq2, fmaf(b1, dq1, q0));
  float hb1 = 0.0f, hb2 = 0.0f, vb1 = 0.0f, vb2 = 0.0f, hq = 1.0f, vq = 1.0f;
  if (!quads) {
    double sxp = (px & 1) ? -1.0 : 1.0, syp = (py & 1) ? -1.0 : 1.0;
    hb1 = (float)fma(sxp, A1, e1) * inv_area, hb2 = (float)fma(sxp, A2, e2) * inv_area;
    vb1 = (float)fma(syp, B1, e1) * inv_area, vb2 = (float)fma(syp, B2, e2) * inv_area;
    hq = fmaf(hb2, dq2, fmaf(hb1, dq1, q0)), vq = fmaf(vb2, dq2, fmaf(vb1, dq1, q0));
  }
  float r, hr = 0.0f, vr = 0.0f;
  if (COMMON && !quads) {
    rcp3_ieee(qq, hq, vq, r, hr, vr);  // one exponent-window test for the three
  } else {
    r = rcp_ieee(qq);
  }
  const uint32_t kind = COMMON ? (uint32_t)PIPE_MESH : ((flags >> F_KIND_SHIFT) & 3u);
  float cr = interp3(s1.z, s3.z, s5.z, b1, b2, r);
  float cg = interp3(s1.w, s3.w, s5.w, b1, b2, r);
  float cb = interp3(s2.x, s4.x, s6.x, b1, b2, r);
  if (kind == PIPE_COLORED_TRIANGLE) return make_float4(cr, cg, cb, 1.0f);  // shaders/colored_triangle.frag:9-12
  float u = interp3(s2.y, s4.y, s6.y, b1, b2, r), v = interp3(s2.z, s4.z, s6.z, b1, b2, r);
  if (!COMMON) {
    hr = rcp_ieee(hq);
    vr = rcp_ieee(vq);
  }
  float uh, vh, uv_, vv_;
  if (quads) {
    uh = quad_partner_x(u), vh = quad_partner_x(v);
    uv_ = quad_partner_y(u), vv_ = quad_partner_y(v);
  } else {
    uh = interp3(s2.y, s4.y, s6.y, hb1, hb2, hr), vh = interp3(s2.z, s4.z, s6.z, hb1, hb2, hr);
    uv_ = interp3(s2.y, s4.y, s6.y, vb1, vb2, vr), vv_ = interp3(s2.z, s4.z, s6.z, vb1, vb2, vr);
  }
  float dudx = (px & 1) ? (u - uh) : (uh - u);
  float dvdx = (px & 1) ? (v - vh) : (vh - v);
  float dudy = (py & 1) ? (u - uv_) : (uv_ - u);
  float dvdy = (py & 1) ? (v - vv_) : (vv_ - v);
  // ---- texture(colorTex, uv): implicit LOD, then one unified 2-level x 4-tap footprint
  float W0 = (float)t.w, H0 = (float)t.h;
  float mx = dudx * W0, my = dvdx * H0;
  float nx_ = dudy * W0, ny_ = dvdy * H0;
  float rho2 = fmaxf(fmaf(mx, mx, my * my), fmaf(nx_, nx_, ny_ * ny_));
  float lambda = fminf(fmaxf(lod_from_rho2(rho2), t.min_lod), t.max_lod);
  const bool linear = COMMON ? true : (((lambda <= 0.0f) ? (t.filters & 1u) : ((t.filters >> 1) & 1u)) != 0u);
  int q = (int)t.levels - 1;
  const bool mip_linear = COMMON ? true : (((t.filters >> 2) & 1u) != 0u);
  int dn = min(max((int)ceilf(lambda + 0.5f) - 1, 0), q);
  float lc = fminf(fmaxf(lambda, 0.0f), (float)q);
  float fl = floorf(lc);
  int dhi = mip_linear ? (int)fl : dn;
  float delta = mip_linear ? lc - fl : 0.0f;
  int dlo = mip_linear ? min(dhi + 1, q) : dn;
  float us = (fabsf(u) < 8388608.0f) ? u : 0.0f, vs = (fabsf(v) < 8388608.0f) ? v : 0.0f;
  // level extents and arena offsets.  The layout pads every level to powers of two (svr_device.h mip_offset), so
  // the next level starts one padded level further on: no second evaluation of the closed form.
  const int whi = (int)max(t.w >> dhi, 1u), hhi = (int)max(t.h >> dhi, 1u);
  const uint32_t bhi = t.base_off + mip_offset(t.lw, t.lh, (uint32_t)dhi);
  Taps th = level_taps<COMMON>(whi, hhi, bhi, linear, us, vs);
  const uint8_t* tb = P.tex_arena;
  uint32_t h00 = texel(tb, th.o00), h10 = texel(tb, th.o10), h01 = texel(tb, th.o01), h11 = texel(tb, th.o11);
  // The second level only matters where delta != 0 (lerp(H, L, 0) == H exactly): magnified and
  // NEAREST-mip pixels skip its four taps when no lane of the wave needs them.  Its loads are issued
  // before anything consumes the first level's texels, so both batches are in flight together.
  const bool two_levels = __any(delta != 0.0f);
  Taps tl = th;
  uint32_t l00 = 0, l10 = 0, l01 = 0, l11 = 0;
  if (two_levels) {
    const int wlo = (int)max(t.w >> dlo, 1u), hlo = (int)max(t.h >> dlo, 1u);
    const uint32_t step = 4u << ((uint32_t)max((int)t.lw - dhi, 0) + (uint32_t)max((int)t.lh - dhi, 0));  // padded bytes of level dhi
    const uint32_t blo = dlo != dhi ? bhi + step : bhi;
    tl = level_taps<COMMON>(wlo, hlo, blo, linear, us, vs);
    l00 = texel(tb, tl.o00);
    l10 = texel(tb, tl.o10);
    l01 = texel(tb, tl.o01);
    l11 = texel(tb, tl.o11);
  }
  float4 tx;
  tx.x = bilerp(h00, h10, h01, h11, 0, th.alpha, th.beta);
  tx.y = bilerp(h00, h10, h01, h11, 1, th.alpha, th.beta);
  tx.z = bilerp(h00, h10, h01, h11, 2, th.alpha, th.beta);
  tx.w = 1.0f;
  if (two_levels) {
    tx.x = lerpf(tx.x, bilerp(l00, l10, l01, l11, 0, tl.alpha, tl.beta), delta);
    tx.y = lerpf(tx.y, bilerp(l00, l10, l01, l11, 1, tl.alpha, tl.beta), delta);
    tx.z = lerpf(tx.z, bilerp(l00, l10, l01, l11, 2, tl.alpha, tl.beta), delta);
  }
  if (TRACE) {  // slots shared with the oracle's trace (tests/tools only)
    trace[0] = (float)(hdr.z >> 2); trace[1] = b1; trace[2] = b2; trace[3] = r; trace[4] = u; trace[5] = v;
    trace[6] = dudx; trace[7] = dvdx; trace[8] = dudy; trace[9] = dvdy; trace[10] = lambda;
    trace[11] = tx.x; trace[12] = tx.y; trace[13] = tx.z;
    trace[26] = hb1; trace[27] = hb2; trace[28] = vb1; trace[29] = vb2; trace[30] = hr; trace[31] = vr;
  }
  if (kind == PIPE_TEX_IMAGE) {  // shaders/tex_image.frag:10-12 — the only consumer of texture alpha
    tx.w = bilerp(h00, h10, h01, h11, 3, th.alpha, th.beta);
    if (two_levels) tx.w = lerpf(tx.w, bilerp(l00, l10, l01, l11, 3, tl.alpha, tl.beta), delta);
    return tx;
  }
  // shaders/mesh.frag:12-19
  float nx = interp3(s0.w, s2.w, s4.w, b1, b2, r);
  float ny = interp3(s1.x, s3.x, s5.x, b1, b2, r);
  float nz = interp3(s1.y, s3.y, s5.y, b1, b2, r);
  const float* L = P.scene.sunlight_direction;
  float d = fmaf(nz, L[2], fmaf(ny, L[1], nx * L[0]));
  float light = fmaxf(d, 0.1f);
  float sunw = P.scene.sunlight_color[3];
  cr = cr * tx.x;
  cg = cg * tx.y;
  cb = cb * tx.z;
  float4 o;
  o.x = fmaf(cr * light, sunw, cr * P.scene.ambient_color[0]);
  o.y = fmaf(cg * light, sunw, cg * P.scene.ambient_color[1]);
  o.z = fmaf(cb * light, sunw, cb * P.scene.ambient_color[2]);
  o.w = 1.0f;
  if (TRACE) {
    trace[15] = nx; trace[16] = ny; trace[17] = nz; trace[18] = cr; trace[19] = cg; trace[20] = cb;
    trace[21] = light; trace[22] = o.x; trace[23] = o.y; trace[24] = o.z; trace[25] = o.w;
  }
  return o;
}

// ------------------------------------------------------------------------------------------------
// colour target codecs: the attachment holds fp16 (reference) or unorm8
template <int FMT>
struct Codec;
template <>
struct Codec<SVR_COLOR_RGBA16F> {
  typedef uint2 enc_t;
  // The fp32 result must exist before it is rounded to fp16 (two roundings, as an attachment store
  // after an fp32 shader does): without the barrier hipcc fuses "fma -> cvt" into v_fma_mixlo_f16,
  // which rounds once and differs from the contract on fp16 ties.
  static __device__ __forceinline__ float pin(float x) {
    asm volatile("" : "+v"(x));
    return x;
  }
  static __device__ __forceinline__ enc_t encode(float4 c) {
    c.x = pin(c.x); c.y = pin(c.y); c.z = pin(c.z); c.w = pin(c.w);
    uint32_t r = __half_as_ushort(__float2half_rn(c.x)), g = __half_as_ushort(__float2half_rn(c.y));
    uint32_t b = __half_as_ushort(__float2half_rn(c.z)), a = __half_as_ushort(__float2half_rn(c.w));
    return make_uint2(r | (g << 16), b | (a << 16));
  }
  static __device__ __forceinline__ float4 decode(enc_t e) {
    return make_float4(__half2float(__ushort_as_half((unsigned short)(e.x & 0xffffu))),
                       __half2float(__ushort_as_half((unsigned short)(e.x >> 16))),
                       __half2float(__ushort_as_half((unsigned short)(e.y & 0xffffu))),
                       __half2float(__ushort_as_half((unsigned short)(e.y >> 16))));
  }
};
template <>
struct Codec<SVR_COLOR_RGBA8> {
  typedef uint32_t enc_t;
  static __device__ __forceinline__ uint32_t un8(float f) {
    return (uint32_t)__float2int_rn(fminf(fmaxf(f, 0.0f), 1.0f) * 255.0f);
  }
  static __device__ __forceinline__ enc_t encode(float4 c) {
    return un8(c.x) | (un8(c.y) << 8) | (un8(c.z) << 16) | (un8(c.w) << 24);
  }
  static __device__ __forceinline__ float4 decode(enc_t e) {
    return make_float4(chan(e, 0), chan(e, 1), chan(e, 2), chan(e, 3));
  }
};

// ------------------------------------------------------------------------------------------------
// Phase A, column scan.  A heavy bin is mostly slivers (a distant column's quads are ~3 x 26 pixels)
// and a whole-wave pass per triangle leaves 95 % of the lanes idle on them.  Instead every triangle of
// a staged batch is expanded into one work item per pixel column of its tile-clamped bbox (wave prefix
// sum over the 64 column counts); one lane takes one item, walks its rows and resolves visibility with
// one LDS ds_max_u64 per covered pixel on a (depth bits << 32 | key) tile.  Lanes are busy whatever
// the triangle shapes are; the four waves take alternate chunks of 64 items.  max over (depth, key)
// is what in-order GREATER_OR_EQUAL with depth write leaves behind, so no order is needed.  The
// record is implied by the key: key - 1 is the triangle's main slot (a clipped parent's slot links to
// its pieces, see resolve_record).
// LIST: the record indices come from s_list (LDS: a quarter's row-filtered copy of the bin) instead of the bin.
template <bool INSTR, bool LIST, bool HIZ>
__device__ __forceinline__ void scan_columns(const FrameParams& P, uint4* s_cov, const uint32_t* s_list, uint32_t bin_base, uint32_t n,
                                             unsigned long long* s_depth, uint32_t* s_bm, int tx0, int ty0, int ry0, int nrows,
                                             uint32_t& n_raster, uint32_t& n_hiz_bad, uint32_t& occl) {
  // rows [ry0, ry0 + nrows) of the tile at (tx0, ty0): the whole tile, or one quarter of a split tile
  auto entry = [&](uint32_t i) -> uint32_t { return LIST ? (s_list[i] & 0x7fffffffu) : P.bins[bin_base + i]; };  // (bit 31: tile_body's filter, instrumented passes)
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  // Staging is double-buffered through registers: the next batch's records (two 16-byte pieces per thread,
  // behind the dependent bin -> record load) are in flight while this batch is walked.  Single-buffered,
  // ~3K of the ~7.6K cycles a batch of 64 triangles costs were this latency, exposed.
  static_assert(BATCH * 8 == 512, "two pieces per thread");
  uint4 pre0 = make_uint4(0, 0, 0, 0), pre1 = pre0;
  {
    uint32_t cnt = min((uint32_t)BATCH, n);
    const bool staged = (threadIdx.x & 7u) < SREC;  // (the eighth piece — texture words — is not staged: SREC)
    if (staged && threadIdx.x < cnt * 8u) pre0 = reinterpret_cast<const uint4*>(P.recs + entry(threadIdx.x >> 3))[threadIdx.x & 7u];
    if (staged && threadIdx.x + 256u < cnt * 8u) pre1 = reinterpret_cast<const uint4*>(P.recs + entry(32u + (threadIdx.x >> 3)))[threadIdx.x & 7u];
  }
  // HIZ — the hierarchical depth test, a choice per pass (tile_kernel: passes whose bins are deep).  Visibility is a
  // maximum over (depth, key), so a triangle whose largest possible depth lies below what EVERY pixel of an 8x8 block
  // already holds cannot change that block: from the second batch on, the smallest depth of each of the tile's sixteen
  // blocks is taken off the visibility tile between two batches (the cells only grow, so the figure stays a lower bound
  // while the batch is walked), and a triangle that can win in none of the blocks it reaches makes no work items.  A
  // deep tile of the 8K x16 frame holds 12 000 triangles in nine layers: nine of ten that arrive are hidden by what is
  // there.  (Testing every column of the survivors against its own blocks as well was measured: the triangle test
  // alone is faster, 8K x16 -8.7 % against -7.9 %.)
  // "Largest possible depth": the largest corner depth plus the rounding the per-pixel chain can add (b1, b2 >= 0 and
  // b1 + b2 <= 1 + 2^-22 inside the triangle, two fma roundings): 2^-21 of the plane's magnitudes is eight times that.
  // s_bm: two sets of sixteen words [bx * 4 + by], filled in turn by LDS atomics: the one being filled was reset
  // while the other was in use, two barriers ago.
  // INSTRUMENTED passes drop nothing (their fragment counts are the oracle's) and CHECK the test instead: every
  // fragment of a triangle it — or tile_body's filter (bit 31 of the list entry) — would have dropped is compared with
  // its cell, and one that wins is counted (Counters::hiz_bad: the host fails the pass).
  constexpr uint32_t SVR_HIZ_MIN = BATCH;      // bins of more than one batch
  constexpr uint32_t SVR_HIZ_KEEP_COLS = 192u; // columns two batches' test has to take out to go on (below)
  const bool hiz = HIZ && n > (uint32_t)SVR_HIZ_MIN;
  uint32_t cur = 0, hz_cols = 0, hz_batches = 0, hz_pause = 0;
  if (hiz && threadIdx.x < 32u) s_bm[threadIdx.x] = threadIdx.x < 16u ? 0u : 0xffffffffu;  // (ordered by the loop's first barrier)
  for (uint32_t b0 = 0; b0 < n; b0 += BATCH) {
    uint32_t cnt = min((uint32_t)BATCH, n - b0);
    __syncthreads();  // previous batch fully consumed
    // The test pays where much of what arrives is hidden, and costs (the sixteen minima, a test per triangle and wave:
    // some 250 wave-instructions per batch, what walking a hundred columns costs) where little is: every two batches the
    // wave looks at how many columns the test took out, and under SVR_HIZ_KEEP_COLS it rests for six batches.
    // Wave-uniform, and the same in all four waves: same records, same block depths.
    const bool test = hiz && b0 && hz_pause == 0u;
    if (hiz && b0 && hz_pause) hz_pause--;
    if (test) {
      cur ^= 1u;
      const uint32_t b = threadIdx.x >> 4, k = threadIdx.x & 15u;  // block (bx = b >> 2, by = b & 3), a thread's four cells of it
      const uint4* c4 = reinterpret_cast<const uint4*>(s_depth + ((b & 3u) * 8u + (k >> 1)) * TILE + (b >> 2) * 8u + (k & 1u) * 4u);
      const uint4 lo = c4[0], hi = c4[1];  // (low word key, high word depth bits) x 4
      atomicMin(&s_bm[cur * 16u + b], min(min(lo.y, lo.w), min(hi.y, hi.w)));  // LDS
      if (threadIdx.x < 16u) s_bm[(cur ^ 1u) * 16u + threadIdx.x] = 0xffffffffu;
    }
    if (threadIdx.x < cnt * 8u && (threadIdx.x & 7u) < SREC) s_cov[(threadIdx.x >> 3) * SREC + (threadIdx.x & 7u)] = pre0;
    if (threadIdx.x + 256u < cnt * 8u && (threadIdx.x & 7u) < SREC) s_cov[(32u + (threadIdx.x >> 3)) * SREC + (threadIdx.x & 7u)] = pre1;
    __syncthreads();
    const uint4* bm4 = reinterpret_cast<const uint4*>(s_bm + cur * 16u);
    if (b0 + BATCH < n) {
      uint32_t nb = b0 + BATCH, ncnt = min((uint32_t)BATCH, n - nb);
      const bool staged = (threadIdx.x & 7u) < SREC;
      if (staged && threadIdx.x < ncnt * 8u) pre0 = reinterpret_cast<const uint4*>(P.recs + entry(nb + (threadIdx.x >> 3)))[threadIdx.x & 7u];
      if (staged && threadIdx.x + 256u < ncnt * 8u) pre1 = reinterpret_cast<const uint4*>(P.recs + entry(nb + 32u + (threadIdx.x >> 3)))[threadIdx.x & 7u];
    }
    // lane i: column count of triangle i inside this tile
    int cx0 = 0, cw = 0;
    bool hidden = false;
    uint32_t gone = 0;  // columns of this lane's triangle the test takes out
    uint32_t zb = 0, claim = 0;  // HIZ: the triangle's largest possible depth; an occluder's smallest over the rectangle (bits)
    if (lane < cnt) {
      uint4 h = s_cov[lane * SREC];
      int minx = (int)(int16_t)(h.x & 0xffffu), miny = (int)(int16_t)(h.x >> 16);
      int maxx = (int)(int16_t)(h.y & 0xffffu), maxy = (int)(int16_t)(h.y >> 16);
      cx0 = max(minx, tx0);
      const int cx1 = min(maxx, tx0 + TILE - 1);
      const int cy0 = max(miny, ry0), cy1 = min(maxy, ry0 + nrows - 1);
      cw = (cx1 >= cx0 && cy1 >= cy0) ? cx1 - cx0 + 1 : 0;
      if (HIZ && cw) {
        const float4 zr = reinterpret_cast<const float4*>(s_cov)[lane * SREC + 1u];
        const float eps = (fabsf(zr.x) + fabsf(zr.y) + fabsf(zr.z)) * 0x1p-21f;
        zb = f2u(fmaxf(fmaxf(zr.x, fmaxf(zr.x + zr.y, zr.x + zr.z)) + eps, 0.0f));  // the largest depth it can have anywhere
        // An OCCLUDER: a triangle that covers every pixel of this workgroup's rows of the tile (a wall or a floor at this
        // resolution covers many tiles) leaves every one of them at least as near as the smallest depth it has over the
        // rectangle, whenever it is walked: whatever cannot reach that depth is hidden in the whole rectangle — also
        // triangles of the same batch and of batches before it (visibility is a maximum: order does not matter).
        // All three edges are >= 0 at the rectangle's four corner pixels (linear: the smallest of the four is C +
        // min(A x0, A x1) + min(B y0, B y1), exact) and the triangle is convex; its depth over the rectangle is at
        // least the plane's smallest corner value, through the same float chain as the walk, less the rounding bound.
        if (minx <= tx0 && maxx >= tx0 + TILE - 1 && miny <= ry0 && maxy >= ry0 + nrows - 1) {
          // (In FLOAT, with the rounding carried as margins — the edge values at the corners need not be exact for a
          // claim that only has to be safe: in fp64, nine coefficients and four corners' chains were more live registers
          // than the fragment stage's peak.  An edge value built from three terms of magnitude M is off by at most
          // 3 * 2^-24 M: "inside" asks for 2^-21 M to spare, and the barycentrics' error, e / area, goes into the depth
          // bound with the plane's slopes.)
          const double* dd = reinterpret_cast<const double*>(s_cov + lane * SREC);  // [4..6] A0 A1 A2, [7..9] B0 B1 B2, [10..12] C0 C1 C2
          const float x0 = (float)tx0, x1 = (float)(tx0 + TILE - 1), y0 = (float)ry0, y1 = (float)(ry0 + nrows - 1);
          bool all_in = true;
          float slack = 0.0f;  // what the float edge values may be off by, as a share of the area: the barycentrics' error
          float bc[2][4];
#pragma unroll
          for (int i = 0; i < 3; i++) {
            const float A = (float)dd[4 + i], B = (float)dd[7 + i], C = (float)dd[10 + i] + ((i == 1 && (h.w & F_T1)) || (i == 2 && (h.w & F_T2)) ? 1.0f : 0.0f);
            const float M = (fabsf(C) + fmaxf(fabsf(A * x0), fabsf(A * x1)) + fmaxf(fabsf(B * y0), fabsf(B * y1))) * 0x1p-21f;
            const float e00 = fmaf(A, x0, fmaf(B, y0, C)), e10 = fmaf(A, x1, fmaf(B, y0, C));
            const float e01 = fmaf(A, x0, fmaf(B, y1, C)), e11 = fmaf(A, x1, fmaf(B, y1, C));
            all_in = all_in && fminf(fminf(e00, e10), fminf(e01, e11)) >= M + 1.0f;  // (+1: the bias of edges 1, 2 was added above; edge 0's is in C)
            if (i) {
              bc[i - 1][0] = e00 * zr.w; bc[i - 1][1] = e10 * zr.w; bc[i - 1][2] = e01 * zr.w; bc[i - 1][3] = e11 * zr.w;
              slack = fmaxf(slack, M * zr.w);
            }
          }
          if (all_in) {
            float zmin = fmaf(bc[1][0], zr.z, fmaf(bc[0][0], zr.y, zr.x));
            zmin = fminf(zmin, fmaf(bc[1][1], zr.z, fmaf(bc[0][1], zr.y, zr.x)));
            zmin = fminf(zmin, fmaf(bc[1][2], zr.z, fmaf(bc[0][2], zr.y, zr.x)));
            zmin = fminf(zmin, fmaf(bc[1][3], zr.z, fmaf(bc[0][3], zr.y, zr.x)));
            zmin -= eps + 2.0f * slack * (fabsf(zr.y) + fabsf(zr.z));
            claim = f2u(fminf(fmaxf(zmin, 0.0f), 1.0f));
          }
        }
      }
    }
    if (HIZ) {  // the nearest occluder so far (wave-uniform, and the same in all four waves: same records)
      unsigned long long claims = __ballot(claim != 0u);
      while (claims) {
        const int l = __ffsll((long long)claims) - 1;
        claims &= claims - 1ull;
        occl = max(occl, (uint32_t)__builtin_amdgcn_readlane((int)claim, l));
      }
      if (!INSTR && cw && zb < occl) cw = 0;  // hidden behind an occluder wherever it reaches
    }
    if (lane < cnt) {
      if (test && cw) {  // the blocks it reaches, against what they hold (the box again from the staged header: kept in
        // registers across the occluders' ballots it is what the allocator spills)
        const uint4 h = s_cov[lane * SREC];
        const int cx1 = min((int)(int16_t)(h.y & 0xffffu), tx0 + TILE - 1);
        const int cy0 = max((int)(int16_t)(h.x >> 16), ry0), cy1 = min((int)(int16_t)(h.y >> 16), ry0 + nrows - 1);
        uint32_t m = 0xffffffffu;
        const int by0 = (cy0 - ty0) >> 3, by1 = (cy1 - ty0) >> 3;
        for (int bx = (cx0 - tx0) >> 3; bx <= (cx1 - tx0) >> 3; bx++) {
          const uint4 q = bm4[bx];
          m = min(m, (by0 <= 0 && by1 >= 0) ? q.x : 0xffffffffu);
          m = min(m, (by0 <= 1 && by1 >= 1) ? q.y : 0xffffffffu);
          m = min(m, (by0 <= 2 && by1 >= 2) ? q.z : 0xffffffffu);
          m = min(m, (by0 <= 3 && by1 >= 3) ? q.w : 0xffffffffu);
        }
        hidden = zb < m;
        if (hidden) gone = (uint32_t)cw;
        if (!INSTR && hidden) cw = 0;  // it can change nothing where it reaches: no work items
      }
      if (INSTR && cw) {  // the verdict (this test's or the filter's tag) for the items' check: bit 31 of the staged flags word; every wave writes the same
        uint32_t* fl = reinterpret_cast<uint32_t*>(s_cov) + lane * SREC * 4u + 3u;
        *fl = (*fl & 0x7fffffffu) | ((hidden || (LIST && (s_list[b0 + lane] >> 31))) ? 0x80000000u : 0u);
      }
    }
    if (test) {
      const uint32_t nh = (uint32_t)__popcll(__ballot(hidden));
      // (a lower bound of the columns taken out, from three ballots: a sum over the lanes would keep a register alive
      // across the scan that the fragment stage lacks)
      hz_cols += nh + 3u * (uint32_t)__popcll(__ballot(gone >= 4u)) + 12u * (uint32_t)__popcll(__ballot(gone >= 16u));
      if (++hz_batches == 2u) {
        if (hz_cols < SVR_HIZ_KEEP_COLS) hz_pause = 6u;
        hz_cols = hz_batches = 0u;
      }
    }
    uint32_t inc = (uint32_t)cw;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t v = __shfl_up(inc, off);
      if ((int)lane >= off) inc += v;
    }
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    // Few columns (a tile under a couple of large triangles): cut every column into 2 or 4 row bands so
    // that all four waves share the walk instead of one wave walking 32 rows per lane.
    const uint32_t sh = total <= 64u ? 2u : (total <= 128u ? 1u : 0u);  // log2(bands)
    const int band_rows = nrows >> sh;
    const uint32_t items = total << sh;
    for (uint32_t c = wave; c * 64u < items; c += 4u) {
      uint32_t item = c * 64u + lane;
      uint32_t j = item >> sh, band = item & ((1u << sh) - 1u);
      bool act = item < items;
      // owner = number of triangles whose inclusive prefix is <= j (binary search over the wave's lanes;
      // a scalar compare chain for bins of <= 8 triangles was tried: no measurable difference)
      uint32_t pos = 0;
#pragma unroll
      for (uint32_t step = 32; step >= 1; step >>= 1) {
        uint32_t v = __shfl(inc, (int)min(pos + step - 1u, 63u));
        if (pos + step <= 64u && v <= j) pos += step;
      }
      uint32_t i = min(pos, 63u);
      uint32_t excl = __shfl(inc, (int)i) - (uint32_t)__shfl(cw, (int)i);
      int col = __shfl(cx0, (int)i) + (int)(j - excl);
      if (!act) continue;
      const uint4* rec = s_cov + i * SREC;
      uint4 h = rec[0];
      int miny = (int)(int16_t)(h.x >> 16), maxy = (int)(int16_t)(h.y >> 16);
      int y0 = max(miny, ry0 + (int)band * band_rows), y1 = min(maxy, ry0 + (int)band * band_rows + band_rows - 1);
      if (y0 > y1) continue;
      uint32_t key = h.z, flags = h.w;
      float4 zr = reinterpret_cast<const float4*>(rec)[1];
      const double2* d = reinterpret_cast<const double2*>(rec);
      double2 c2 = d[2], c3 = d[3], c4 = d[4], c5 = d[5], c6 = d[6];
      double B0 = c3.y, B1 = c4.x, B2 = c4.y;
      double u1 = (flags & F_T1) ? 1.0 : 0.0, u2 = (flags & F_T2) ? 1.0 : 0.0;
      double dx = (double)col, dy = (double)y0;
      double f0 = fma(c2.x, dx, fma(B0, dy, c5.x));
      double f1 = fma(c2.y, dx, fma(B1, dy, c5.y));
      double f2 = fma(c3.x, dx, fma(B2, dy, c6.x));
      unsigned long long* cell = s_depth + (y0 - ty0) * TILE + (col - tx0);
      for (int y = y0; y <= y1; y++) {
        if (f0 >= 0.0 && f1 >= 0.0 && f2 >= 0.0) {
          if (INSTR) n_raster++;
          float b1 = (float)(f1 + u1) * zr.w, b2 = (float)(f2 + u2) * zr.w;
          float z = fmaf(b2, zr.z, fmaf(b1, zr.y, zr.x));
          z = fminf(fmaxf(z, 0.0f), 1.0f) + 0.0f;
          if (INSTR && (flags >> 31) && f2u(z) >= (uint32_t)(*cell >> 32)) n_hiz_bad++;  // (cells only grow: no false alarm)
          atomicMax(cell, ((unsigned long long)f2u(z) << 32) | key);
        }
        f0 += B0;  // exact: integers below 2^53
        f1 += B1;
        f2 += B2;
        cell += TILE;
      }
    }
  }
}

// (key >> 2) - 1 is the main record slot, and for keys with bit 0 clear that is the record.  Bit 0 set:
// the triangle went through the clipper, its slot is an invalid record that links to the contiguous
// block of its pieces (k_bin.hip clip_and_bin); exactly one of them covers the pixel (they
// partition the parent under the top-left rule).  The flag keeps this dependent load out of the
// common path.
__device__ __forceinline__ uint32_t resolve_record(const FrameParams& P, uint32_t rec, int px, int py) {
  uint4 h = *reinterpret_cast<const uint4*>(P.recs + rec);
  if ((int)(int16_t)(h.x & 0xffffu) <= (int)(int16_t)(h.y & 0xffffu)) return rec;
  double dx = (double)px, dy = (double)py;
  for (uint32_t c = 0; c < h.w; c++) {
    const TriRec* t = P.recs + h.z + c;
    if (t->minx > t->maxx || px < t->minx || px > t->maxx || py < t->miny || py > t->maxy) continue;
    if (fma(t->A[0], dx, fma(t->B[0], dy, t->C[0])) >= 0.0 && fma(t->A[1], dx, fma(t->B[1], dy, t->C[1])) >= 0.0 &&
        fma(t->A[2], dx, fma(t->B[2], dy, t->C[2])) >= 0.0)
      return h.z + c;
  }
  return rec;  // unreachable for a pixel that produced a fragment
}

// ------------------------------------------------------------------------------------------------
// Transparent pass, ordered form.  Blending is order dependent (the target rounds after every
// blend), so fragments must reach each pixel in submission order.  The tile's transparent bin is
// sorted by submission key once (rank_sort below), then scanned once with the same (triangle, column)
// items as phase A.  For the order to survive, wave w owns the tile's
// rows 8w..8w+7: it takes the items of every triangle that touches its band, in bin order, and all
// its lanes step through the SAME absolute row at the same time — so the fragments of one pixel are
// appended to the wave's LDS queue in submission order (__ballot + prefix popcount: lane order is
// item order).  Depth-passing fragments are shaded 64 at a time by whichever lanes are free and
// blended into the wave's LDS colour band.  Several fragments of one pixel can sit in the same group
// of 64: the lowest lane of every pixel applies them in lane order = queue order (flush_fragments).
// In a quarter of a split tile (svr_device.h SPLIT_*) a wave owns 2 rows instead of 8.
constexpr uint32_t SORT_CAP = SPLIT_SORT_MAX;             // (1392) bins above this are sorted in the global sort arena instead of LDS
constexpr uint32_t RANK_SORT_MAX = 1024;                  // bins up to this are ranked on their 32-bit keys (rank_sort), up to SORT_CAP by rank_sort_big
#ifndef SVR_HIZ_FILTER_MIN
#define SVR_HIZ_FILTER_MIN 1024
#endif
constexpr uint32_t HIZ_AVG_ENTRIES = 96;                  // passes whose bins hold this many entries per tile on average run phase A with the hierarchical depth test
constexpr uint32_t HIZ_FILTER_MIN = SVR_HIZ_FILTER_MIN;                 // opaque bins above this go through the hidden-triangle filter (tile_body) window by window
constexpr uint32_t QUARTER_LIST_CAP = 3808;               // entries of a quarter's row-filtered opaque list (LDS behind the depth tile); bins are taken in windows of this
constexpr uint32_t QUEUE_CAP = 128;                       // < 64 carried over + up to 64 new per row step

template <int FMT, bool INSTR>
__device__ __forceinline__ void flush_fragments(const FrameParams& P, typename Codec<FMT>::enc_t* col, uint2* q,
                                                float4* s_src, uint32_t& qn, int tx0, int by0,
                                                uint32_t lane, uint32_t& n_shaded) {
  typedef Codec<FMT> CD;
  uint32_t cnt = min(qn, 64u);
  bool act = lane < cnt;
  uint2 e = q[act ? lane : 0u];
  uint32_t pix = e.x, rec = e.y;  // pix = row in band * 32 + column in tile
  int px = tx0 + (int)(pix & 31u), py = by0 + (int)(pix >> 5);
  float4 src = make_float4(0.f, 0.f, 0.f, 0.f);
  // the specialised fragment stage (mesh.frag, trilinear sampler, power-of-two image: the key's common-case bit rides
  // in the queued record index) when every fragment of the group carries it, as in phase B: the curtain tiles' ordered
  // pass — 36 layers deep, the kernel's slowest tiles — is nine tenths shading
  if (__all(!act || (rec & REC_COMMON))) {  // wave-uniform
    if (act) src = shade_pixel<false, true>(P, rec & ~REC_COMMON, px, py, nullptr);
  } else if (act) {
    src = shade_pixel<false>(P, rec & ~REC_COMMON, px, py, nullptr);
  }
  if (INSTR && act) n_shaded++;
  // Blending is ordered per pixel (C13) and a group of 64 queued fragments often holds many layers of
  // the same few pixels (a curtain seen edge-on).  Every pixel's fragments are collected as a lane mask,
  // and the lowest lane of each mask blends its pixel's fragments in lane order = submission order, in
  // registers: one LDS read per layer instead of an election round per layer.  The masks come from eight
  // ballots, one per bit of the pixel's number in the wave's band (pix < 256): a lane keeps, bit by bit, the lanes
  // that agree with it — some fifty instructions per group of 64 fragments whatever it holds.  (A 64-bit word
  // per pixel of the band in LDS, filled by atomics, was 8 KiB of the workgroup's 37: the difference between four
  // and five workgroups per CU.  One ballot per DISTINCT pixel was tried first: up to 64 rounds per group, as
  // long as the fragment stage itself.)
  if (act) s_src[lane] = src;
  unsigned long long mm = __ballot(act);
#pragma unroll
  for (uint32_t b = 0; b < 8u; b++) {
    const bool bit = (pix >> b) & 1u;
    const unsigned long long with = __ballot(bit);
    mm &= bit ? with : ~with;
  }
  mm = act ? mm : 0ull;
  __builtin_amdgcn_wave_barrier();  // LDS operations of a wave retire in order; this pins the compiler's order too
  if (act && (mm & ((1ull << lane) - 1ull)) == 0ull) {
    float4 dst = CD::decode(col[pix]);
    typename CD::enc_t out = col[pix];
    while (mm) {
      uint32_t j = (uint32_t)__ffsll((long long)mm) - 1u;
      mm &= mm - 1ull;
      float4 sj = s_src[j];
      // enable_blending_additive: rgb = src*ONE + dst*DST_ALPHA, a = src*ONE + dst*ZERO
      float4 o = make_float4(fmaf(dst.x, dst.w, sj.x), fmaf(dst.y, dst.w, sj.y), fmaf(dst.z, dst.w, sj.z), sj.w);
      if (INSTR && P.trace_buf && px == P.trace_x && py == P.trace_y) {
        (void)shade_pixel<true>(P, q[j].y & ~REC_COMMON, px, py, P.trace_buf);
        float* tb = P.trace_buf;
        tb[32] = dst.x; tb[33] = dst.y; tb[34] = dst.z; tb[35] = dst.w;
        tb[36] = o.x; tb[37] = o.y; tb[38] = o.z; tb[39] = o.w;
      }
      out = CD::encode(o);
      dst = CD::decode(out);  // the attachment holds the rounded value between layers
    }
    col[pix] = out;
  }
  uint32_t rest = qn - cnt;
  for (uint32_t j = lane; j < rest; j += 64u) {  // lock-step: every read of a step precedes its writes
    uint2 v = q[cnt + j];
    q[j] = v;
  }
  qn = rest;
}

// The thread's index in the workgroup for the phases behind visibility, from the wave's number (an SGPR: wv) and
// the lane's: three instructions where they are wanted.  threadIdx.x itself arrives in v0 and cannot be made again;
// read in these phases it stays live from the top of the kernel through the fragment stage, and the allocator answers by
// spilling it — the kernel then carries a scratch frame, which costs every dispatch ~2.5 us (tools/gapbench.hip).
// volatile asm: copies must not be merged back into one long-lived value.
__device__ __forceinline__ uint32_t tid_of(uint32_t wv) {
  uint32_t lane;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
  return (wv << 6) | lane;
}

// Workgroup-wide OR of a per-thread flag: every wave posts its own vote, one barrier, everybody reads the four.
// (HIP's __syncthreads_or / _and go through the device library's workgroup reduction, which forms a three-
// dimensional thread index out of v0 — and so keeps v0 alive, spilled, across the whole kernel.)  SLOT: votes that
// may follow each other without a barrier in between must use different slots.
template <int SLOT>
__device__ __forceinline__ bool block_any(bool v, uint32_t wv) {
  __shared__ __attribute__((aligned(16))) uint32_t votes[2][4];
  votes[SLOT][wv] = __any(v) ? 1u : 0u;  // 64 lanes, one word, one value
  __syncthreads();
  const uint4 all = *reinterpret_cast<const uint4*>(votes[SLOT]);
  return (all.x | all.y | all.z | all.w) != 0u;
}

template <int FMT, bool INSTR>
__device__ __forceinline__ void scan_columns_ordered(const FrameParams& P, uint4* s_cov, uint32_t* s_idx, const uint32_t* order,
                                                     uint32_t n, int tx0, int ty0, int row0, uint32_t lrpw, const uint32_t* s_z,
                                                     typename Codec<FMT>::enc_t* col, uint2* q, float4* s_src,
                                                     uint32_t& n_raster, uint32_t& n_shaded, const uint32_t wv) {
  const uint32_t tid = tid_of(wv), lane = tid & 63u, wave = wv;
  const unsigned long long below = (1ull << lane) - 1ull;
  const int rpw = 1 << lrpw;  // rows per wave: 8, or 2 in a quarter
  const int by0 = ty0 + row0 + rpw * (int)wave, by1 = by0 + rpw - 1;  // this wave's rows
  uint32_t qn = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += BATCH) {
    uint32_t cnt = min((uint32_t)BATCH, n - b0);
    __syncthreads();
    {  // a thread's two pieces: both indices first, then both records — two round trips per batch, not four
      // (the thread index afresh per batch: the staging addresses made from it are otherwise hoisted out of the loop,
      // where they are the values that get spilled)
      const uint32_t st = tid_of(wv), t0 = st >> 3, pc = st & 7u;
      uint32_t ri0 = 0, ri1 = 0;
      if (t0 < cnt) ri0 = order[b0 + t0];
      if (t0 + 32u < cnt) ri1 = order[b0 + 32u + t0];
      uint4 r0 = make_uint4(0, 0, 0, 0), r1 = r0;
      if (t0 < cnt && pc < SREC) r0 = reinterpret_cast<const uint4*>(P.recs + ri0)[pc];
      if (t0 + 32u < cnt && pc < SREC) r1 = reinterpret_cast<const uint4*>(P.recs + ri1)[pc];
      if (t0 < cnt) {
        if (pc < SREC) s_cov[t0 * SREC + pc] = r0;
        if (pc == 0) s_idx[t0] = ri0;
      }
      if (t0 + 32u < cnt) {
        if (pc < SREC) s_cov[(t0 + 32u) * SREC + pc] = r1;
        if (pc == 0) s_idx[t0 + 32u] = ri1;
      }
    }
    __syncthreads();
    int cx0 = 0, cw = 0;
    if (lane < cnt) {
      uint4 h = s_cov[lane * SREC];
      int minx = (int)(int16_t)(h.x & 0xffffu), miny = (int)(int16_t)(h.x >> 16);
      int maxx = (int)(int16_t)(h.y & 0xffffu), maxy = (int)(int16_t)(h.y >> 16);
      cx0 = max(minx, tx0);
      int cx1 = min(maxx, tx0 + TILE - 1);
      cw = (cx1 >= cx0 && min(maxy, by1) >= max(miny, by0)) ? cx1 - cx0 + 1 : 0;
    }
    uint32_t inc = (uint32_t)cw;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t v = __shfl_up(inc, off);
      if ((int)lane >= off) inc += v;
    }
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    // The scan is RESUMABLE: it runs until the wave's queue holds a full group of 64 fragments, stops, the group is
    // shaded, and the scan takes up again at (chunk c, row t) — a wave-uniform pair; everything else about a lane's
    // item is made again from the staged records.  Shading inside the row loop kept the whole scan state (edge
    // values and increments in fp64, the z plane, the item's coordinates: some forty registers) alive across the
    // inlined fragment stage: 124 VGPRs, four waves per SIMD, for a phase 1.5 % of the tiles run.
    uint32_t c = 0;
    int t = 0;
    for (;;) {
      bool full = false;
      while (c * 64u < total) {  // every chunk, in order: items are sorted by submission key
        uint32_t j = c * 64u + lane;
        bool act = j < total;
        uint32_t pos = 0;
#pragma unroll
        for (uint32_t step = 32; step >= 1; step >>= 1) {
          uint32_t v = __shfl(inc, (int)min(pos + step - 1u, 63u));
          if (pos + step <= 64u && v <= j) pos += step;
        }
        uint32_t i = min(pos, 63u);
        uint32_t excl = __shfl(inc, (int)i) - (uint32_t)__shfl(cw, (int)i);
        int colx = __shfl(cx0, (int)i) + (int)(j - excl) - tx0;  // column inside the tile
        colx = act ? colx : 0;
        const uint4* rec = s_cov + i * SREC;
        uint4 h = rec[0];
        int y0 = (int)(int16_t)(h.x >> 16), y1 = (int)(int16_t)(h.y >> 16);
        uint32_t flags = h.w, ri = s_idx[i];
        float4 zr = reinterpret_cast<const float4*>(rec)[1];
        const double2* d = reinterpret_cast<const double2*>(rec);
        double2 c2 = d[2], c3 = d[3], c4 = d[4], c5 = d[5], c6 = d[6];
        double B0 = c3.y, B1 = c4.x, B2 = c4.y;
        double dx = (double)(tx0 + colx), dy = (double)(by0 + t);  // (exact integers below 2^53: the same values the row-by-row additions reach)
        double f0 = fma(c2.x, dx, fma(B0, dy, c5.x));
        // edges 1 and 2 are walked UNBIASED (g = f + 1 where C carries the top-left bias, else f): that is the value the
        // barycentrics want, and for integers "f >= 0" is "g > 0" there and "g >= 0" elsewhere — two compares where
        // holding the two 1.0 / 0.0 in registers across walk and flush cost four VGPRs and an add per row
        // (walking edges 1 and 2 unbiased, with "g > 0" / "g >= 0" picked per lane, saves the two adds and four
        // registers below but pays more in the compares' mask logic: +4 % on the curtain tiles)
        const double u1 = (flags & F_T1) ? 1.0 : 0.0, u2 = (flags & F_T2) ? 1.0 : 0.0;
        double g1 = fma(c2.y, dx, fma(B1, dy, c5.y));
        double g2 = fma(c3.x, dx, fma(B2, dy, c6.x));
#pragma unroll 1
        for (; t < rpw; t++) {  // the same absolute row by0 + t in every lane
          int y = by0 + t;
          bool inside = act && y >= y0 && y <= y1 && f0 >= 0.0 && g1 >= 0.0 && g2 >= 0.0;
          if (INSTR) n_raster += inside ? 1u : 0u;
          float b1 = (float)(g1 + u1) * zr.w, b2 = (float)(g2 + u2) * zr.w;
          float z = fmaf(b2, zr.z, fmaf(b1, zr.y, zr.x));
          z = fminf(fmaxf(z, 0.0f), 1.0f) + 0.0f;
          bool pass = inside && f2u(z) >= s_z[(y - ty0) * TILE + colx];  // GREATER_OR_EQUAL vs opaque depth, no write
          unsigned long long m = __ballot(pass);
          f0 += B0;
          g1 += B1;
          g2 += B2;
          if (m) {
            if (pass) q[qn + (uint32_t)__popcll(m & below)] = make_uint2((uint32_t)(t * TILE + colx), ri | ((h.z & 2u) ? REC_COMMON : 0u));
            qn += (uint32_t)__popcll(m);
            if (qn >= 64u) {  // stop here: the group is shaded with none of the above alive
              full = true;
              t++;
              break;
            }
          }
        }
        if (full) break;
        t = 0;
        c++;
      }
      if (!full) break;  // the batch is exhausted
      flush_fragments<FMT, INSTR>(P, col, q, s_src, qn, tx0, by0, lane, n_shaded);
      if (t == rpw) {
        t = 0;
        c++;
      }
    }
  }
  while (qn) flush_fragments<FMT, INSTR>(P, col, q, s_src, qn, tx0, by0, lane, n_shaded);
}

// Rank by counting: with unique keys the number of smaller keys IS the sorted position.  Every lane reads the
// same key per step (an LDS broadcast) against its own <= 8; a handful of barriers in all, where the bitonic
// network below pays one per compare-exchange step (55 of them at 1024 entries: 75K cycles for the
// 590-triangle curtain bins).
// The 32-bit submission keys alone are unique unless the clipper's pieces of one triangle share a tile;
// the rank loop runs on them (a 64-bit compare issues at half rate), every element then claims its rank
// in an LDS table, and a claim that did not stick means a tie: the (rare) bin is ranked again on
// (key, record index) words, which are always unique — the order must not depend on which workgroup
// sorts, the quarters of a split tile all write the same list.
template <uint32_t K, typename KeyT>
__device__ __forceinline__ void count_ranks(const KeyT* s, uint32_t n, const KeyT (&mine)[K], uint32_t (&rank)[K]) {
#pragma unroll
  for (uint32_t k = 0; k < K; k++) rank[k] = 0;
  for (uint32_t j = 0; j < n; j++) {
    KeyT v = s[j];
#pragma unroll
    for (uint32_t k = 0; k < K; k++) rank[k] += v < mine[k] ? 1u : 0u;
  }
}

template <uint32_t K>
__device__ __forceinline__ void rank_sort(const FrameParams& P, unsigned char* lds, uint32_t bin_base, uint32_t n, uint32_t* out, const uint32_t wv) {
  const uint32_t tid = tid_of(wv);
  uint32_t* k32 = reinterpret_cast<uint32_t*>(lds);         // [n] keys
  uint32_t* claim = k32 + SORT_CAP;                          // [n] element that owns each rank
  uint32_t ri[K], key[K], rank[K];
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = tid + 256u * k;
    ri[k] = i < n ? P.bins[bin_base + i] : 0u;
  }
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = tid + 256u * k;
    key[k] = i < n ? P.recs[ri[k]].key : 0u;
    if (i < n) k32[i] = key[k];
  }
  __syncthreads();
  count_ranks<K, uint32_t>(k32, n, key, rank);
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = tid + 256u * k;
    if (i < n) claim[rank[k]] = i;
  }
  __syncthreads();
  bool lost = false;
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = tid + 256u * k;
    lost = lost || (i < n && claim[rank[k]] != i);
  }
  if (block_any<0>(lost, wv)) {  // equal keys in the bin: rank (key, record index) instead
    unsigned long long* k64 = reinterpret_cast<unsigned long long*>(lds);
    unsigned long long wide[K];
#pragma unroll
    for (uint32_t k = 0; k < K; k++) {
      uint32_t i = tid + 256u * k;
      wide[k] = ((unsigned long long)key[k] << 32) | ri[k];
      if (i < n) k64[i] = wide[k];
    }
    __syncthreads();
    count_ranks<K, unsigned long long>(k64, n, wide, rank);
  }
#pragma unroll
  for (uint32_t k = 0; k < K; k++)
    if (tid + 256u * k < n) out[rank[k]] = ri[k];
}

template <uint32_t K>
__device__ __forceinline__ void rank_pass(const uint32_t* k32, const uint32_t* r32, uint32_t* marks, uint32_t base, uint32_t n, uint32_t* out, const uint32_t wv) {
  const uint32_t tid = tid_of(wv);
  uint32_t mine[K], rank[K];
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = base + tid + 256u * k;
    mine[k] = i < n ? k32[i] : 0xffffffffu;
  }
  count_ranks<K, uint32_t>(k32, n, mine, rank);
#pragma unroll
  for (uint32_t k = 0; k < K; k++) {
    uint32_t i = base + tid + 256u * k;
    if (i < n) {
      out[rank[k]] = r32[i];
      marks[rank[k]] = 1u;
    }
  }
}

// Bins of 1025 .. SORT_CAP entries: the same rank by counting with keys AND record indices parked in LDS, every thread
// taking its eight elements four at a time (eight keys, records and ranks per thread at once were the register
// peak of the whole kernel, paid for by spills in the phases every tile runs).  Ties (the clipper's pieces of one
// triangle sharing the tile) leave a rank unclaimed; such a bin is ranked again on (key, position in the bin) —
// pieces of one triangle never cover the same pixel, so their mutual order is free, and every workgroup that
// sorts this bin (the quarters of a split tile) reads it in the same order.
// marks: [SORT_CAP] words of LDS outside the scratch block (the record staging buffer, idle during the sort).
__device__ __forceinline__ void rank_sort_big(const FrameParams& P, unsigned char* lds, uint32_t* marks, uint32_t bin_base, uint32_t n,
                                              uint32_t* out, const uint32_t wv) {
  const uint32_t tid = tid_of(wv);
  uint32_t* k32 = reinterpret_cast<uint32_t*>(lds);  // [n] keys
  uint32_t* r32 = k32 + SORT_CAP;                     // [n] record indices
  {  // all record indices, then all keys: two round trips for the whole bin
    uint32_t ri[8], key[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      uint32_t i = tid + 256u * k;
      ri[k] = i < n ? P.bins[bin_base + i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      uint32_t i = tid + 256u * k;
      key[k] = i < n ? P.recs[ri[k]].key : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      uint32_t i = tid + 256u * k;
      if (i < n) {
        k32[i] = key[k];
        r32[i] = ri[k];
        marks[i] = 0u;
      }
    }
  }
  __syncthreads();  // every entry of the bin has been read: sorting in place is safe
  rank_pass<4>(k32, r32, marks, 0u, n, out, wv);
  switch ((n - 1024u + 255u) >> 8) {  // the elements from 1024 on: as many per thread as there are
    case 1: rank_pass<1>(k32, r32, marks, 1024u, n, out, wv); break;
    case 2: rank_pass<2>(k32, r32, marks, 1024u, n, out, wv); break;
    case 3: rank_pass<3>(k32, r32, marks, 1024u, n, out, wv); break;
    default: rank_pass<4>(k32, r32, marks, 1024u, n, out, wv); break;
  }
  __syncthreads();
  bool hole = false;
  for (uint32_t r = tid; r < n; r += 256u) hole = hole || marks[r] == 0u;
  if (block_any<0>(hole, wv)) {  // equal keys: rank (key, position)
    for (uint32_t i = tid; i < n; i += 256u) {
      const uint32_t m = k32[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < n; j++) {
        uint32_t v = k32[j];
        rank += (v < m || (v == m && j < i)) ? 1u : 0u;
      }
      out[rank] = r32[i];
    }
  }
}

// A quarter of a split tile makes its OWN sorted list of the tile's transparent bin: only the triangles whose
// bounding box meets the quarter's eight rows (and the tile's columns) are kept, ranked by key among themselves and
// written to the quarter's span of the sort arena.  What it leaves out has no coverage in the quarter, so the ordered
// scan sees the same fragments in the same order.  (Every quarter used to sort the whole bin and walk all of it: at
// 1920x1080 the quarter of a tile that sees a curtain edge-on IS the frame — 1224 triangles, 232 K cycles in phase
// C of which ~55 K the rank by counting, whose cost goes with the square of the list.)
// lds: >= 8 * SORT_CAP bytes (keys, record indices); marks: [SORT_CAP] words elsewhere in LDS.  Returns the length of the list.
__device__ __forceinline__ uint32_t sort_quarter_bin(const FrameParams& P, unsigned char* lds, uint32_t* marks, uint32_t bin_base, uint32_t n,
                                                     int tx0, int qy0, int qrows, uint32_t* out, const uint32_t wv) {
  __shared__ uint32_t kept;
  const uint32_t tid = tid_of(wv);
  uint32_t* k32 = reinterpret_cast<uint32_t*>(lds);  // [m] keys
  uint32_t* r32 = k32 + SORT_CAP;                     // [m] record indices
  if (tid == 0) kept = 0u;
  __syncthreads();
  {
    uint32_t ri[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      uint32_t i = tid + 256u * k;
      ri[k] = i < n ? P.bins[bin_base + i] : 0u;
    }
    uint2 box[8];
    uint32_t key[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      uint32_t i = tid + 256u * k;
      if (i < n) {
        box[k] = *reinterpret_cast<const uint2*>(P.recs + ri[k]);
        key[k] = P.recs[ri[k]].key;
      } else {
        box[k] = make_uint2(1u, 0u);  // minx 1 > maxx 0
        key[k] = 0u;
      }
    }
    const uint32_t lane = tid & 63u;
#pragma unroll
    for (uint32_t k = 0; k < 8; k++) {
      if (256u * k >= n) break;  // uniform
      const int minx = (int)(int16_t)(box[k].x & 0xffffu), miny = (int)(int16_t)(box[k].x >> 16);
      const int maxx = (int)(int16_t)(box[k].y & 0xffffu), maxy = (int)(int16_t)(box[k].y >> 16);
      const bool keep = min(maxx, tx0 + TILE - 1) >= max(minx, tx0) && min(maxy, qy0 + qrows - 1) >= max(miny, qy0);
      const unsigned long long m = __ballot(keep);
      uint32_t base = 0;
      if (lane == 0 && m) base = atomicAdd(&kept, (uint32_t)__popcll(m));  // LDS
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (keep) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        k32[pos] = key[k];
        r32[pos] = ri[k];
        marks[pos] = 0u;
      }
    }
  }
  __syncthreads();
  const uint32_t m = kept;
  if (m == 0u) return 0u;
  for (uint32_t base = 0; base < m; base += 1024u) {  // four elements per thread at a time
    switch (min((m - base + 255u) >> 8, 4u)) {
      case 1: rank_pass<1>(k32, r32, marks, base, m, out, wv); break;
      case 2: rank_pass<2>(k32, r32, marks, base, m, out, wv); break;
      case 3: rank_pass<3>(k32, r32, marks, base, m, out, wv); break;
      default: rank_pass<4>(k32, r32, marks, base, m, out, wv); break;
    }
  }
  __syncthreads();
  bool hole = false;
  for (uint32_t r = tid; r < m; r += 256u) hole = hole || marks[r] == 0u;
  if (block_any<0>(hole, wv)) {  // equal keys (the clipper's pieces of one triangle): rank (key, record) — their mutual order is free
    for (uint32_t i = tid; i < m; i += 256u) {
      const uint32_t mk = k32[i], mr = r32[i];
      uint32_t rank = 0;
      for (uint32_t j = 0; j < m; j++) {
        uint32_t v = k32[j];
        rank += (v < mk || (v == mk && r32[j] < mr)) ? 1u : 0u;
      }
      out[rank] = mr;
    }
  }
  __threadfence_block();
  __syncthreads();
  return m;
}

// s: scratch — the LDS block (>= 8 * SORT_CAP bytes) for bins up to SORT_CAP, else the tile's span of the global sort arena,
// the next power of two >= n words (fill_kernel reserved it; global memory is coherent inside a workgroup's CU).
// out: where the sorted record indices go — the bin itself, or (quarters of a split tile, n <= RANK_SORT_MAX) the
// tile's words of the sort arena, which all four quarters fill with the same values.
// WIDE: bins of 1025 .. SORT_CAP entries keep all their (up to six) elements of a thread in registers (rank_sort<5..8>) instead of
// going through rank_sort_big.  For the kernel built with the quarter path (passes of up to 4096 tiles: 1080p, the
// row bands of a multi-GPU run) whose frame time IS its deepest curtain tile: 2.5 % of the 1080p frame; in the
// whole-tile kernel the same choice costs the phases every tile runs more (registers) than it returns.
template <bool WIDE>
__device__ __forceinline__ void sort_bin_by_key(const FrameParams& P, unsigned long long* s, uint32_t* marks, uint32_t bin_base, uint32_t n,
                                                uint32_t* out, const uint32_t wv) {
  const uint32_t tid = tid_of(wv);
  if (n <= SORT_CAP) {
    unsigned char* lds = reinterpret_cast<unsigned char*>(s);
    switch ((n + 255u) >> 8) {
      case 0:
      case 1: rank_sort<1>(P, lds, bin_base, n, out, wv); break;
      case 2: rank_sort<2>(P, lds, bin_base, n, out, wv); break;
      case 3: rank_sort<3>(P, lds, bin_base, n, out, wv); break;
      case 4: rank_sort<4>(P, lds, bin_base, n, out, wv); break;
      default:
        if (WIDE) {
          switch ((n + 255u) >> 8) {
            case 5: rank_sort<5>(P, lds, bin_base, n, out, wv); break;
            case 6: rank_sort<6>(P, lds, bin_base, n, out, wv); break;
            case 7: rank_sort<7>(P, lds, bin_base, n, out, wv); break;
            default: rank_sort<8>(P, lds, bin_base, n, out, wv); break;
          }
        } else {
          rank_sort_big(P, lds, marks, bin_base, n, out, wv);
        }
        break;
    }
    __threadfence_block();
    __syncthreads();
    return;
  }
  // bitonic sort of the bin's (key << 32 | record) words by all 256 threads
  uint32_t np = 64;
  while (np < n) np <<= 1;
  for (uint32_t i = tid; i < np; i += 256u) {
    unsigned long long v = ~0ull;
    if (i < n) {
      uint32_t ri = P.bins[bin_base + i];
      v = ((unsigned long long)P.recs[ri].key << 32) | ri;
    }
    s[i] = v;
  }
  __syncthreads();
  for (uint32_t k = 2; k <= np; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < np; i += 256u) {
        uint32_t x = i ^ j;
        if (x > i) {
          unsigned long long a = s[i], b = s[x];
          bool asc = (i & k) == 0;
          if ((a > b) == asc) {
            s[i] = b;
            s[x] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (uint32_t i = tid; i < n; i += 256u) out[i] = (uint32_t)s[i];
  __threadfence_block();
  __syncthreads();
}

// LDS block of a tile workgroup (s_c), by phase:
//   [0, 8K)    phase A: the (depth << 32 | key) tile.  From phase B on: the tile's colour, row-major, in the
//              target's encoding — shaded pixels go straight there instead of living in registers until the
//              write-back (with depth, keys and colours of four pixels per lane in VGPRs the kernel sat on the
//              128-register line and every change to the fragment stage paid in scratch traffic)
//   [8K, 12K)  the tile's opaque depth bits for phase C's depth test (phase A of a quarter: its triangle list, to the end)
//   [12K, 23K) phase C: sort scratch (11 KiB), then per wave its fragment queue | shaded colours (2 KiB each)
constexpr uint32_t LDS_Z_OFF = TILE * TILE * 8;
constexpr uint32_t LDS_C_OFF = LDS_Z_OFF + TILE * TILE * 4;
constexpr uint32_t WAVE_C_BYTES = QUEUE_CAP * 8 + 64 * 16;  // queue | shaded colours
// the four waves' blocks, or the sort's scratch (keys + record indices of SORT_CAP entries) before the scan starts.
// The workgroup's LDS — this + the 8 KiB staging buffer + a few words — stays under 32 000 bytes (25 granules of 1280 B):
// five workgroups per CU.
constexpr uint32_t PHASE_C_BYTES = LDS_C_OFF + (SORT_CAP * 8 > 4 * WAVE_C_BYTES ? SORT_CAP * 8 : 4 * WAVE_C_BYTES);

// The lane's pixel (rx, ry) in the tile and its word li in the tile's LDS images, made afresh from the thread index
// where a later phase needs them.  Kept in registers from the top of the kernel they are the two values the allocator
// spills, and a kernel with a scratch frame pays for it at every dispatch (2.5 us per launch, tools/gapbench.hip).
__device__ __forceinline__ void lane_pixel(uint32_t wv, int& rx, int& ry, uint32_t& li) {
  const uint32_t tid = tid_of(wv);
  rx = (int)(((tid >> 6) & 1u) * 16u + (tid & 7u));
  ry = (int)((tid >> 7) * 16u + ((tid >> 3) & 7u));
  li = (uint32_t)ry * TILE + (uint32_t)rx;
}

// A 16-byte piece of a target row leaving the tile.  Non-temporal: the targets are written once per pass and not
// read by it, and 100 MB of them per 4K frame otherwise sit dirty in the L2s until the end-of-kernel write-back and
// push records and texels out (-1.5 % of the frame; the same hint on the geometry stage's record stores costs 0.7 %:
// those are read back within the pass).
__device__ __forceinline__ void store_row16(void* p, uint4 v) {
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x4*>(p));
}

// QUARTER: this workgroup renders 8 rows of a split tile (svr_device.h SPLIT_*).  Its own instantiation, chosen
// by blockIdx alone: sharing one body with run-time row ranges cost the whole-tile path 20-35 spilled
// registers and 8-13 % of the frame, and choosing by a flag in tile_info put a dependent load in front of
// every tile (+2 %).
template <int FMT, bool INSTR, bool QUARTER, bool SPLIT>
__device__ __forceinline__ void tile_body(const FrameParams& P, const uint4 i0, const uint4 i1, uint4* s_cov, uint32_t* s_idx, unsigned char* s_c,
                                          const uint32_t wv, const bool hiz_on, const uint32_t wg_start = 0) {
  typedef Codec<FMT> CD;
  typedef typename CD::enc_t enc_t;
  // Workgroups are dispatched in blockIdx order: walk the tiles heaviest class first (fill_kernel's
  // tile_order).  Tiles are dealt round-robin over the 8 XCDs; a contiguous span per XCD was tried
  // and loses: the heavy rows of the frame all land on one XCD and the other seven idle.
  // i0, i1: the launch slot's 32 bytes of tile_info (the tile and its two bins, written by fill_kernel in launch order).
  uint32_t tile, n_op, n_tr, off_op, off_tr, sort_base = 0, rows = 3u << 8, is_split = 0;
  if (P.tuning & TUNE_NO_TILE_ORDER) {
    tile = blockIdx.x;
    n_op = P.tile_count[tile];
    n_tr = P.tile_count[P.n_tiles + tile];
    off_op = P.tile_offset[tile];
    off_tr = P.tile_offset[P.n_tiles + tile];
  } else {
    tile = i0.x; n_op = i0.y; off_op = i0.z; n_tr = i0.w; off_tr = i1.x; sort_base = i1.y; rows = i1.z; is_split = i1.w;
  }
  if (SPLIT && !QUARTER && is_split) return;  // a split tile's slot in the ordinary launch order: its quarters head the launch
#ifndef SVR_AB_NO_PRIO
  // The kernel lasts as long as its slowest tiles: the ones that see a curtain edge-on (hundreds of transparent triangles,
  // ~35 layers: 36 000 fragments to shade on 1024 pixels, in order) start first and are still running when every other
  // tile is done — all the longer the more workgroups share their SIMDs (five per CU: resident 167 us of the kernel's 168).
  // Their waves ask for issue priority: what they take from the neighbours, the neighbours have to spare.
  if (tile_cost(n_op, n_tr) >= SVR_PRIO_COST) __builtin_amdgcn_s_setprio(3);
#endif
  // this workgroup's rows of the tile, row0 .. row0 + nrows - 1; pixels outside them are treated like pixels outside the scissor
  const int row0 = QUARTER ? (int)(rows & 0xffu) : 0;
  constexpr uint32_t lrpw = QUARTER ? 1u : 3u;  // log2(rows per wave in phase C)
  constexpr int nrows = 4 << lrpw;
  const uint32_t tx = tile % P.tiles_x, ty = tile / P.tiles_x;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const int tx0 = (int)(P.sx + tx * TILE), ty0 = tile_row_y(P, (int)ty);
  const int x_end = (int)(P.sx + P.sw), y_end = (int)(P.sy + P.sh);
  const int sub_y0 = ty0 + row0;
  // pixel ownership: wave w the 16x16 quadrant w, lane l one pixel in each of its four 8x8 blocks; slot k of a
  // lane is the pixel (rx + (k & 1) * 8, ry + (k >> 1) * 8) of the tile, word li + (k & 1) * 8 + (k >> 1) * 256 of its LDS images
  const int rx = (int)((wave & 1u) * 16u + (lane & 7u)), ry = (int)((wave >> 1) * 16u + (lane >> 3));
  const uint32_t li = (uint32_t)ry * TILE + (uint32_t)rx;
  // the whole tile lies inside the scissor (every pixel of it is this workgroup's to write): wave-uniform
  const bool inside = !QUARTER && tx0 + TILE <= x_end && ty0 + TILE <= y_end;
  const bool aligned = ((P.W | P.sx) & 3u) == 0u;

  bool pix_ok[4];
  uint32_t recs[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    int px = tx0 + rx + (k & 1) * 8, py = ty0 + ry + (k >> 1) * 8;
    pix_ok[k] = px < x_end && py < y_end && (!QUARTER || (uint32_t)(py - sub_y0) < (uint32_t)nrows);
    recs[k] = NO_REC;
  }
  uint32_t n_raster = 0, n_shaded = 0, n_hiz_bad = 0;
  uint32_t occl = 0;  // scan_columns<HIZ>: depth bits every pixel of this workgroup's rows is claimed to reach (an occluder's smallest)
  long long stamp[5] = {0, 0, 0, 0, 0};  // SVR_OPT_TILE_CYCLES: shader-clock stamps per phase
  const bool stamps = P.tile_cycles != nullptr;
  if (stamps) stamp[0] = clock64();

  unsigned long long* s_depth = reinterpret_cast<unsigned long long*>(s_c);
  enc_t* lc = reinterpret_cast<enc_t*>(s_c);
  uint32_t* s_z = reinterpret_cast<uint32_t*>(s_c + LDS_Z_OFF);

  // ---- phase A: opaque visibility
  if (n_op) {
    for (uint32_t i = threadIdx.x; i < TILE * TILE; i += 256u) s_depth[i] = 0ull;  // ordered by scan_columns' first barrier
    const bool hiz_filter = hiz_on && n_op > HIZ_FILTER_MIN;
    if ((QUARTER && n_op > 2u * BATCH) || hiz_filter) {
      // A quarter of an opaque-heavy tile: most of the bin's triangles do not reach its 8 rows, and staging
      // them costs as much as in the whole tile.  One pass over the record headers (indices, then bounding
      // rows: two round trips per 1024 entries) leaves the quarter's own list in LDS, behind the depth tile.
      // The bin is taken in windows of QUARTER_LIST_CAP entries (what the list holds if all of a window stay):
      // visibility is a maximum, so it may be found window by window.  (Bins beyond the capacity used to be
      // walked whole by each of the four quarters: configs[4]'s deepest tiles hold 12 000 triangles, and a
      // rank of eight was as slow as those tiles' quarters — 0.44 ms for an eighth of a 1.5-ms tile stage.)
      // DEEP bins (more than HIZ_FILTER_MIN entries), of whole tiles too: the same pass reads the record's depth plane
      // with its box and drops what is hidden — the hierarchical depth test of scan_columns, before a record is staged
      // (there a hidden triangle still costs its place in a batch: two barriers and a round trip per 64, and in the
      // 8K x16 frame nine of ten are hidden).  Windows of 1024 entries then, the sixteen block depths taken off the
      // visibility tile between two of them.  Instrumented passes keep everything and tag what would go (bit 31).
      uint32_t* s_list = reinterpret_cast<uint32_t*>(s_c + LDS_Z_OFF);
      uint32_t* f_bm = s_idx + 48;  // [16] block depths of the filter (scan_columns has s_idx[16..47])
      const uint32_t WIN = hiz_filter ? 1024u : QUARTER_LIST_CAP;
      for (uint32_t win = 0; win < n_op; win += WIN) {
        const uint32_t n_win = min(n_op - win, WIN);
        const bool ftest = hiz_filter && win != 0u;
        __syncthreads();  // the previous window's list is consumed, its count read by everybody
        if (threadIdx.x == 0) s_idx[0] = 0u;
        if (ftest && threadIdx.x < 16u) f_bm[threadIdx.x] = 0xffffffffu;
        __syncthreads();
        if (ftest) {
          const uint32_t b = threadIdx.x >> 4, k = threadIdx.x & 15u;  // block (bx = b >> 2, by = b & 3), a thread's four cells of it
          const uint4* c4 = reinterpret_cast<const uint4*>(s_depth + ((b & 3u) * 8u + (k >> 1)) * TILE + (b >> 2) * 8u + (k & 1u) * 4u);
          const uint4 lo = c4[0], hi = c4[1];
          atomicMin(&f_bm[b], min(min(lo.y, lo.w), min(hi.y, hi.w)));  // LDS
          __syncthreads();
        }
        for (uint32_t base = 0; base < n_win; base += 1024u) {
          uint32_t ri[4];
          uint2 box[4];
          float4 zr[4];
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            uint32_t i = base + 256u * k + threadIdx.x;
            ri[k] = i < n_win ? P.bins[off_op + win + i] : 0u;
          }
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            uint32_t i = base + 256u * k + threadIdx.x;
            box[k] = i < n_win ? *reinterpret_cast<const uint2*>(P.recs + ri[k]) : make_uint2(0u, 0u);
            zr[k] = (ftest && i < n_win) ? reinterpret_cast<const float4*>(P.recs + ri[k])[1] : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            uint32_t i = base + 256u * k + threadIdx.x;
            int minx = (int)(int16_t)(box[k].x & 0xffffu), miny = (int)(int16_t)(box[k].x >> 16);
            int maxx = (int)(int16_t)(box[k].y & 0xffffu), maxy = (int)(int16_t)(box[k].y >> 16);
            bool keep = i < n_win && maxy >= sub_y0 && miny <= sub_y0 + nrows - 1;
            uint32_t tag = 0u;
            if (ftest && keep) {
              const int cx0 = max(minx, tx0), cx1 = min(maxx, tx0 + TILE - 1);
              const int cy0 = max(miny, sub_y0), cy1 = min(maxy, sub_y0 + nrows - 1);
              const float zmax = fmaxf(zr[k].x, fmaxf(zr[k].x + zr[k].y, zr[k].x + zr[k].z)) + (fabsf(zr[k].x) + fabsf(zr[k].y) + fabsf(zr[k].z)) * 0x1p-21f;
              const uint32_t zb = f2u(fmaxf(zmax, 0.0f));
              uint32_t m = 0xffffffffu;
              const int by0 = (cy0 - ty0) >> 3, by1 = (cy1 - ty0) >> 3;
              const uint4* bm4 = reinterpret_cast<const uint4*>(f_bm);
              for (int bx = (cx0 - tx0) >> 3; bx <= (cx1 - tx0) >> 3; bx++) {
                const uint4 q = bm4[bx];
                m = min(m, (by0 <= 0 && by1 >= 0) ? q.x : 0xffffffffu);
                m = min(m, (by0 <= 1 && by1 >= 1) ? q.y : 0xffffffffu);
                m = min(m, (by0 <= 2 && by1 >= 2) ? q.z : 0xffffffffu);
                m = min(m, (by0 <= 3 && by1 >= 3) ? q.w : 0xffffffffu);
              }
              if (cx1 >= cx0 && (zb < m || zb < occl)) {  // (occl: the nearest occluder claim of the windows before, scan_columns)
                if (INSTR) tag = 0x80000000u;
                else keep = false;
              }
            }
            unsigned long long m = __ballot(keep);
            uint32_t at = 0;
            if (lane == 0 && m) at = atomicAdd(&s_idx[0], (uint32_t)__popcll(m));  // LDS
            at = __shfl(at, 0);
            if (keep) s_list[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = ri[k] | tag;
          }
        }
        __syncthreads();
        const uint32_t n_mine = s_idx[0];
        if (hiz_on) scan_columns<INSTR, true, true>(P, s_cov, s_list, 0u, n_mine, s_depth, s_idx + 16, tx0, ty0, sub_y0, nrows, n_raster, n_hiz_bad, occl);
        else scan_columns<INSTR, true, false>(P, s_cov, s_list, 0u, n_mine, s_depth, s_idx + 16, tx0, ty0, sub_y0, nrows, n_raster, n_hiz_bad, occl);
      }
    } else {
      if (hiz_on) scan_columns<INSTR, false, true>(P, s_cov, nullptr, off_op, n_op, s_depth, s_idx + 16, tx0, ty0, QUARTER ? sub_y0 : ty0, nrows, n_raster, n_hiz_bad, occl);
      else scan_columns<INSTR, false, false>(P, s_cov, nullptr, off_op, n_op, s_depth, s_idx + 16, tx0, ty0, QUARTER ? sub_y0 : ty0, nrows, n_raster, n_hiz_bad, occl);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; k++) {  // the winners' records move into the owning lanes' registers
      unsigned long long v = s_depth[li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256)];
      // (instrumented passes drop nothing: every pixel of the workgroup's rows must then have reached what the
      // occluders of scan_columns claimed for it — the claim everything dropped behind them rests on)
      if (INSTR && occl && (!QUARTER || (uint32_t)(ry + (k >> 1) * 8 - row0) < (uint32_t)nrows) && (uint32_t)(v >> 32) < occl) n_hiz_bad++;
      if ((uint32_t)v != 0u) {
        uint32_t main_slot = ((uint32_t)v >> 2) - 1u;
        uint32_t rec = ((uint32_t)v & 1u) ? resolve_record(P, main_slot, tx0 + rx + (k & 1) * 8, ty0 + ry + (k >> 1) * 8) : main_slot;
        recs[k] = rec | (((uint32_t)v & 2u) ? REC_COMMON : 0u);
      }
    }
  }
  // The depth target is final here (transparent fragments test but do not write, src/vk_engine.cpp:1673-1674):
  // it leaves now, straight from the visibility tile, as whole rows where the tile lies inside the scissor (16
  // bytes per lane, full 128-byte lines), and phase C's copy of the opaque depth is taken on the way.
  if (inside && aligned) {
    const uint32_t tid = tid_of(wv), row = tid >> 3, c = (tid & 7u) * 4u;
    uint4 z = make_uint4(0u, 0u, 0u, 0u);  // depth CLEAR 0.0
    if (n_op) {
      const uint4* src = reinterpret_cast<const uint4*>(s_depth + row * TILE + c);
      uint4 lo = src[0], hi = src[1];
      z = make_uint4(lo.y, lo.w, hi.y, hi.w);
    }
    store_row16(P.depth + (size_t)(ty0 + (int)row) * P.W + (size_t)(tx0 + (int)c), z);
    if (n_tr) *reinterpret_cast<uint4*>(s_z + row * TILE + c) = z;
  } else {
    int rx, ry;
    uint32_t li;
    lane_pixel(wv, rx, ry, li);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const uint32_t w = li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256);
      uint32_t z = n_op ? (uint32_t)(s_depth[w] >> 32) : 0u;
      if (pix_ok[k]) P.depth[(size_t)(ty0 + ry + (k >> 1) * 8) * P.W + (size_t)(tx0 + rx + (k & 1) * 8)] = u2f(z);
      if (n_tr) s_z[w] = z;
    }
  }
  __syncthreads();  // the visibility tile is dead: its memory is the tile's colour from here on

  if (stamps) stamp[1] = clock64();
  // ---- phase B: shade visible pixels once, into the LDS colour tile
  bool dirty[4];
#pragma unroll
  for (int k = 0; k < 4; k++) dirty[k] = recs[k] != NO_REC;
  uint32_t generic_slots = 0;  // wave-uniform: pixel slots in which some lane needs the generic fragment stage
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint32_t tid = tid_of(wv);
    const int qx = (int)(((tid >> 6) & 1u) * 16u + (tid & 7u)) + (k & 1) * 8, qy = (int)((tid >> 7) * 16u + ((tid >> 3) & 7u)) + (k >> 1) * 8;
    const int px = tx0 + qx, py = ty0 + qy;
    if (__all(!dirty[k] || (recs[k] & REC_COMMON))) {
      // every quad of this 8x8 block of one triangle (or empty): lane ^ 1 is the pixel to the left or right, lane ^ 8 above or below
      const int rk = (int)recs[k];
      const bool quads = __all(!dirty[k] || (__builtin_amdgcn_mov_dpp(rk, 0xB1, 0xf, 0xf, true) == rk && __builtin_amdgcn_mov_dpp(rk, 0x128, 0xf, 0xf, true) == rk));
      if (dirty[k]) {
        lc[(uint32_t)qy * TILE + (uint32_t)qx] = CD::encode(shade_pixel<false, true>(P, recs[k] & ~REC_COMMON, px, py, nullptr, quads));
        if (INSTR) {
          n_shaded++;
          if (P.trace_buf && px == P.trace_x && py == P.trace_y) (void)shade_pixel<true>(P, recs[k] & ~REC_COMMON, px, py, P.trace_buf);
        }
      }
    } else {
      generic_slots |= 1u << k;
    }
  }
  if (generic_slots) {
    // Anything but the common case goes through ONE rolled instance of the generic stage (k is a run-time
    // value, the per-slot state is picked with selects): four unrolled copies of it in this kernel cost
    // the common path registers and instruction-cache room for nothing.
#pragma unroll 1
    for (int k = 0; k < 4; k++) {
      if (!((generic_slots >> k) & 1u)) continue;
      uint32_t rec = (k == 0 ? recs[0] : (k == 1 ? recs[1] : (k == 2 ? recs[2] : recs[3]))) & ~REC_COMMON;
      bool d = k == 0 ? dirty[0] : (k == 1 ? dirty[1] : (k == 2 ? dirty[2] : dirty[3]));
      int rx, ry;
      uint32_t li;
      lane_pixel(wv, rx, ry, li);
      int px = tx0 + rx + (k & 1) * 8, py = ty0 + ry + (k >> 1) * 8;
      if (d) {
        lc[li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256)] = CD::encode(shade_pixel<false>(P, rec, px, py, nullptr));
        if (INSTR) {
          n_shaded++;
          if (P.trace_buf && px == P.trace_x && py == P.trace_y) (void)shade_pixel<true>(P, rec, px, py, P.trace_buf);
        }
      }
    }
  }

  if (P.lazy_clear) {  // the deferred svr_clear_color: every pixel the opaque geometry left untouched
    enc_t cv;
    if constexpr (sizeof(enc_t) == 8) cv = enc_t(make_uint2(P.clear_lo, P.clear_hi));
    else cv = enc_t(P.clear_lo);
    int rx, ry;
    uint32_t li;
    lane_pixel(wv, rx, ry, li);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (!dirty[k] && pix_ok[k]) {
        lc[li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256)] = cv;
        dirty[k] = true;
      }
  }
  if (stamps) stamp[2] = clock64();
  // ---- phase C: transparent fragments in submission order
  if (n_tr) {
    uint32_t tbase = off_tr;
    const uint32_t* order = P.bins + tbase;
    unsigned long long* scratch = reinterpret_cast<unsigned long long*>(s_c + LDS_C_OFF);
    uint32_t n_list = n_tr;
    if (QUARTER) {  // n_tr <= SORT_CAP: the quarter's own part of the bin, sorted in LDS, written to its span of the arena
      uint32_t* own_list = reinterpret_cast<uint32_t*>(P.sort_arena + sort_base + (uint32_t)(row0 >> 3) * ((n_tr + 1u) >> 1));
      n_list = sort_quarter_bin(P, reinterpret_cast<unsigned char*>(scratch), reinterpret_cast<uint32_t*>(s_cov), tbase, n_tr, tx0, sub_y0, nrows, own_list, wv);
      order = own_list;
    } else if (n_tr <= SORT_CAP) {
      sort_bin_by_key<SPLIT>(P, scratch, reinterpret_cast<uint32_t*>(s_cov), tbase, n_tr, P.bins + tbase, wv);
    } else {
      sort_bin_by_key<SPLIT>(P, P.sort_arena + sort_base, reinterpret_cast<uint32_t*>(s_cov), tbase, n_tr, P.bins + tbase, wv);  // rare: a bin too large for LDS
    }
    if (n_list) {  // (a quarter none of the bin's triangles reaches leaves its pixels as they are)
    int rx, ry;
    uint32_t li;
    lane_pixel(wv, rx, ry, li);
#pragma unroll
    for (int k = 0; k < 4; k++)  // colour loadOp LOAD for what neither the opaque pass nor a clear has written
      if (!dirty[k] && pix_ok[k])
        lc[li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256)] =
            reinterpret_cast<const enc_t*>(P.color)[(size_t)(ty0 + ry + (k >> 1) * 8) * P.W + (size_t)(tx0 + rx + (k & 1) * 8)];
    unsigned char* mine = s_c + LDS_C_OFF + wv * WAVE_C_BYTES;
    enc_t* col = lc + (uint32_t)(row0 + ((int)wv << lrpw)) * TILE;  // this wave's rows of the colour tile
    uint2* q = reinterpret_cast<uint2*>(mine);
    float4* s_src = reinterpret_cast<float4*>(mine + QUEUE_CAP * 8);
    scan_columns_ordered<FMT, INSTR>(P, s_cov, s_idx, order, n_list, tx0, ty0, row0, lrpw, s_z, col, q, s_src, n_raster, n_shaded, wv);
#pragma unroll
    for (int k = 0; k < 4; k++) dirty[k] = pix_ok[k];
    }
  }

  if (stamps) stamp[3] = clock64();
  // ---- phase D: the colour tile goes out (the barrier retires phase B's and phase C's writes to it)
  // A tile that lies inside the scissor and has every pixel written leaves as whole rows: 16 bytes per
  // lane, every 128-byte line of the tile's rows written by one instruction.  (A lane's own pixels are
  // 8-pixel row pieces of four 8x8 blocks: stored directly they reach memory as 32- and 64-byte partial
  // lines, which the memory side counted as twice the bytes.)
  const bool whole = dirty[0] && dirty[1] && dirty[2] && dirty[3];
  if (!block_any<1>(!whole, wv) && inside && aligned) {
    constexpr uint32_t PX = 16u / sizeof(enc_t);  // pixels per 16-byte store
#pragma unroll
    for (uint32_t i = tid_of(wv); i < TILE * TILE / PX; i += 256u) {
      uint32_t row = i / (TILE / PX), c = (i % (TILE / PX)) * PX;
      uint4 v = *reinterpret_cast<const uint4*>(lc + row * TILE + c);
      store_row16(reinterpret_cast<enc_t*>(P.color) + (size_t)(ty0 + (int)row) * P.W + (size_t)(tx0 + (int)c), v);
    }
  } else {
    int rx, ry;
    uint32_t li;
    lane_pixel(wv, rx, ry, li);
#pragma unroll
    for (int k = 0; k < 4; k++)
      if (pix_ok[k] && dirty[k])
        reinterpret_cast<enc_t*>(P.color)[(size_t)(ty0 + ry + (k >> 1) * 8) * P.W + (size_t)(tx0 + rx + (k & 1) * 8)] =
            lc[li + (uint32_t)((k & 1) * 8 + (k >> 1) * 256)];
  }
  if (stamps) {
    stamp[4] = clock64();
    if (tid_of(wv) == 0 && row0 == 0) {  // of a split tile: its first quarter
#ifdef SVR_DEBUG_WG_TIMES  // development build (tools/frames.py --wgtimes): when and where the workgroup ran, not its phases
      P.tile_cycles[tile * 4u + 0] = wg_start;
      P.tile_cycles[tile * 4u + 1] = (uint32_t)wall_clock64();
      P.tile_cycles[tile * 4u + 2] = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_ID
      P.tile_cycles[tile * 4u + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // XCC_ID
#else
      for (int k = 0; k < 4; k++) P.tile_cycles[tile * 4u + k] = (uint32_t)(stamp[k + 1] - stamp[k]);
#endif
    }
  }
  if (INSTR) {
    for (int off = 32; off > 0; off >>= 1) {
      n_raster += __shfl_down(n_raster, off);
      n_shaded += __shfl_down(n_shaded, off);
    }
    if (lane == 0) {
      atomicAdd(&P.counters->rasterized, (unsigned long long)n_raster);
      atomicAdd(&P.counters->shaded, (unsigned long long)n_shaded);
    }
    if (__any(n_hiz_bad != 0u) && n_hiz_bad) atomicAdd(&P.counters->hiz_bad, n_hiz_bad);
  }
}

// SPLIT: the launch is headed by SPLIT_EXTRA slots for the quarters of split tiles.  A kernel of its own: the
// quarter path merely compiled in costs the whole-tile path 2-3 % (registers, code size), which a pass with
// more than SPLIT_TILES_MAX tiles — where no tile is worth splitting — need not pay.
#ifndef SVR_TILE_WAVES
#define SVR_TILE_WAVES 5  // waves per SIMD (= workgroups per CU) the tile kernel is compiled for (A/B builds: tools/build_variant.sh)
#endif
template <int FMT, bool INSTR, bool SPLIT>
__global__ __launch_bounds__(256, SVR_TILE_WAVES) void tile_kernel(FrameParams P) {
  __shared__ uint4 s_cov[BATCH * 8];
  __shared__ uint32_t s_idx[BATCH];
  __shared__ __attribute__((aligned(16))) unsigned char s_c[PHASE_C_BYTES];  // phase A depth tile, phase C blocks
  static_assert(LDS_Z_OFF + QUARTER_LIST_CAP * 4 <= PHASE_C_BYTES, "depth tile + a quarter's triangle list");
  static_assert(SPLIT_SORT_MAX <= SORT_CAP && PHASE_C_BYTES - LDS_C_OFF >= SORT_CAP * 8 && PHASE_C_BYTES - LDS_C_OFF >= RANK_SORT_MAX * 8 &&
                    PHASE_C_BYTES - LDS_C_OFF >= 4 * WAVE_C_BYTES, "sort scratch aliases the waves' phase-C blocks");
  // gfx950 hands out its 160 KiB of LDS in granules of 1280 bytes (320 dwords): five workgroups per CU need 25 granules
  // each, 32 000 bytes — NOT 32 768 (at 32 064 bytes the kernel stayed at four per CU: tools/frames.py --wgtimes)
  static_assert(sizeof(s_cov) + sizeof(s_idx) + PHASE_C_BYTES + 64 <= 25 * 1280, "five workgroups per CU: 25 LDS granules of 1280 B");
  static_assert(TILE * TILE * 8 >= TILE * TILE * sizeof(uint2), "the colour tile aliases the visibility tile");

  // Everything the workgroup needs before it can start comes in ONE round of scalar loads: the failure flags
  // and the launch slot's tile_info.  (Written as plain loads they compiled to a chain of four round trips
  // — flags, branch, tile id by a vector load, the other fields by a second one — in front of every tile's
  // bin -> record chain: ~2 K of a light tile's 33 K cycles.)  Constant address space = scalar loads; all of
  // it was written by earlier kernels.
#ifdef SVR_DEBUG_WG_TIMES
  const uint32_t wg_start = (uint32_t)wall_clock64();
#else
  const uint32_t wg_start = 0;
#endif
  const uint32_t slot = SPLIT ? blockIdx.x : SPLIT_EXTRA + blockIdx.x;
  typedef const __attribute__((address_space(4))) uint32_t* const_words;
  const_words ti = (const_words)(const void*)P.tile_info + 8u * slot;
  const_words cnt = (const_words)(const void*)P.counters;
  uint32_t w0 = ti[0], w1 = ti[1], w2 = ti[2], w3 = ti[3], w4 = ti[4], w5 = ti[5], w6 = ti[6], w7 = ti[7];
  uint32_t overflow = cnt[offsetof(Counters, overflow) / 4], n_split = cnt[offsetof(Counters, n_split) / 4];
  uint32_t entries = cnt[offsetof(Counters, total_entries) / 4];
  uint32_t poison = *(const_words)(const void*)P.poison;
  // (pins all of the loads in front of the first branch: left alone, the compiler sinks the tile_info loads
  // behind the flags' round trip)
  asm volatile("" : "+s"(w0), "+s"(w1), "+s"(w2), "+s"(w3), "+s"(w4), "+s"(w5), "+s"(w6), "+s"(w7), "+s"(overflow), "+s"(poison), "+s"(n_split), "+s"(entries));
  const uint4 i0 = make_uint4(w0, w1, w2, w3), i1 = make_uint4(w4, w5, w6, w7);
  const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // this wave's number, for tid_of()
  // The hierarchical depth test of phase A (scan_columns, tile_body's filter) is a choice per PASS: it pays where bins are
  // deep — the 8K x16 frame: 204 entries per tile on average, nine of ten triangles of its deep bins hidden: -10.5 % —
  // and costs 1.25 % of the 4K frame (44 per tile), where what it drops carries 4 % of the fragments.  The bins'
  // total is the pass's own (offsets_kernel), so the choice is the same for every workgroup.
  const bool hiz_on = !(P.tuning & TUNE_NO_HIZ) && ((P.tuning & TUNE_HIZ) || entries >= HIZ_AVG_ENTRIES * P.n_tiles);

  if (blockIdx.x == 0)  // the pass's cost per tile row, for the host (nobody waits for it)
    for (uint32_t r = threadIdx.x; r < min(P.tiles_y, ROW_COST_MAX); r += blockDim.x)
      __hip_atomic_store(P.host_row_cost + r, P.row_cost[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // A pass that overflowed a queue is void, and so is everything after it until the host has replayed
  // it (svr_api.hip "the operation log"): the targets stay as they were before the failed pass.
  if (overflow | poison) {
    if (blockIdx.x == 0 && poison == 0u) {  // the first failure: flag + tell the host which pass, and by how much
      // its counters go along (they keep counting past the capacities), so the replay can size the queues
      // before its first attempt; the set's device copy may be zeroed by a later pass's prologue by then
      if (threadIdx.x < sizeof(Counters) / 4)
        __hip_atomic_store(reinterpret_cast<uint32_t*>(P.host_counters) + threadIdx.x,
                           reinterpret_cast<const uint32_t*>(P.counters)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (threadIdx.x == 0) {
        *P.poison = 1u;
        __hip_atomic_store(P.host_failed_seq, P.op_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
  } else if (SPLIT) {
    if (blockIdx.x < SPLIT_EXTRA) {  // the quarters of split tiles, as many as fill_kernel made
      if (blockIdx.x >= 4u * min(n_split, SPLIT_MAX)) return;
      tile_body<FMT, INSTR, true, true>(P, i0, i1, s_cov, s_idx, s_c, wv, hiz_on, wg_start);
    } else {
      tile_body<FMT, INSTR, false, true>(P, i0, i1, s_cov, s_idx, s_c, wv, hiz_on, wg_start);
    }
  } else {
    tile_body<FMT, INSTR, false, false>(P, i0, i1, s_cov, s_idx, s_c, wv, hiz_on, wg_start);
  }
}

// Instrumented passes only: the counters go to the host (pinned, device-visible) by a one-wave kernel
// behind the tile kernel.  (A D2H copy packet there costs ~15 us of stream time; a last-workgroup-
// reports epilogue in the tile kernel holds every workgroup's slot for an atomic round trip: +15 %.)
// Other passes report nothing (the host learns of an overflow through host_failed_seq), except
// device-flattened ones, whose draw / triangle / culled counts only exist on the device.
__global__ __launch_bounds__(64) void report_kernel(FrameParams P) {
  if (threadIdx.x < sizeof(Counters) / 4)
    __hip_atomic_store(reinterpret_cast<uint32_t*>(P.host_counters) + threadIdx.x,
                       reinterpret_cast<const uint32_t*>(P.counters)[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// start (may be null): signalled when the tile kernel starts.  done: signalled when the pass's last kernel has finished.  It rides on that kernel's
// own dispatch packet (hipExtLaunchKernel's stopEvent): a separate hipEventRecord is one more packet
// for the command processor between two tile kernels.
void launch_tiles(const FrameParams& P, int color_format, bool count_fragments, hipStream_t s, hipEvent_t start, hipEvent_t done) {
  const bool split = !(P.tuning & (TUNE_NO_SPLIT | TUNE_NO_TILE_ORDER));
  dim3 grid(split ? P.n_tiles + SPLIT_EXTRA : P.n_tiles), block(256);
  const bool report = count_fragments || P.flatten;
  hipEvent_t tile_done = report ? nullptr : done;
#ifdef SVR_DEBUG_LDS_PAD  // development builds only (tools/build_variant.sh): extra dynamic LDS per workgroup from the
  // environment, i.e. fewer workgroups per CU without a rebuild; the product reads no environment variable
  static const uint32_t lds_pad = [] { const char* e = getenv("SVR_TILE_LDS_PAD"); return e ? (uint32_t)atoi(e) : 0u; }();
#else
  constexpr uint32_t lds_pad = 0u;
#endif
  // Workgroups per CU.  Both instances fit five (<= 96 VGPRs, 25 of gfx950's 1280-byte LDS granules).  A pass of more
  // than SPLIT_TILES_MAX tiles is tile-bound — stage 1 of the next pass has slack and takes the slots retiring workgroups
  // free — and runs at five: 8K x16 1.512 -> 1.425 ms per frame.  Small passes — a 1080p frame, a rank's share of a
  // sharded one — are bounded by STAGE 1: there a fifth tile workgroup per CU takes the room stage 1 of the next pass runs
  // in (1080p 0.091 -> 0.100 ms, a rank of eight at 8K 0.367 -> 0.384), so their launches claim one LDS granule more and
  // stay at four.  (A/B builds, tools/build_variant.sh: SVR_AB_TILE_PAD pads every launch, SVR_AB_NO_PAD none.)
#if defined(SVR_AB_TILE_PAD)
  const uint32_t pad = 1280u;
#elif defined(SVR_AB_NO_PAD)
  const uint32_t pad = 0u;
#else
  const uint32_t pad = P.n_tiles <= SPLIT_TILES_MAX ? 1280u : 0u;
#endif
#define SVR_LAUNCH_TILES(FMT, INSTR, SPLIT) hipExtLaunchKernelGGL((tile_kernel<FMT, INSTR, SPLIT>), grid, block, lds_pad + pad, s, start, tile_done, 0, P)
  if (color_format == SVR_COLOR_RGBA16F) {
    if (count_fragments) {
      if (split) SVR_LAUNCH_TILES(SVR_COLOR_RGBA16F, true, true);
      else SVR_LAUNCH_TILES(SVR_COLOR_RGBA16F, true, false);
    } else {
      if (split) SVR_LAUNCH_TILES(SVR_COLOR_RGBA16F, false, true);
      else SVR_LAUNCH_TILES(SVR_COLOR_RGBA16F, false, false);
    }
  } else {
    if (count_fragments) {
      if (split) SVR_LAUNCH_TILES(SVR_COLOR_RGBA8, true, true);
      else SVR_LAUNCH_TILES(SVR_COLOR_RGBA8, true, false);
    } else {
      if (split) SVR_LAUNCH_TILES(SVR_COLOR_RGBA8, false, true);
      else SVR_LAUNCH_TILES(SVR_COLOR_RGBA8, false, false);
    }
  }
#undef SVR_LAUNCH_TILES
  if (report) hipExtLaunchKernelGGL(report_kernel, dim3(1), dim3(64), 0, s, nullptr, done, 0, P);
}

}  // namespace svr
