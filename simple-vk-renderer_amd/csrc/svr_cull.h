// svr_cull.h — is_visible (src/vk_engine.cpp:56-86) for the host side of svr_draw_geometry.  Host code only.
// glm 0.9.9 scalar operation order (glm is an unpinned submodule of the reference), no fma:
// mat4*mat4 column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3;  mat4*vec4 = (m0*v0 + m1*v1) + (m2*v2 + m3*v3).
// tests/test_cull_cpu.py checks it against the oracle's scalar restatement on millions of random boxes.
#pragma once

#include <cstring>

#include "../../include/svr.h"

namespace svr {

// Four matrix rows per operation (vector extensions; every lane does the scalar code's operations in its order, no
// fma, so the verdicts are the oracle's bit for bit): the cull of a hundred objects was 10 of the 47 us of host time per
// pass that bound a rank of an eight-way split frame.
typedef float v4f __attribute__((vector_size(16)));
typedef int v4i __attribute__((vector_size(16)));
inline v4f splat4(float x) { return v4f{x, x, x, x}; }
inline v4f load4(const float* p) {
  v4f v;
  std::memcpy(&v, p, 16);
  return v;
}
inline v4f select4(v4i mask, v4f a, v4f b) {  // mask lane all-ones: a, zero: b
  v4i ai, bi;
  std::memcpy(&ai, &a, 16);
  std::memcpy(&bi, &b, 16);
  v4i r = (ai & mask) | (bi & ~mask);
  v4f out;
  std::memcpy(&out, &r, 16);
  return out;
}
inline bool is_visible(const SvrRenderObject& obj, const float* viewproj) {
  static const float corners[8][3] = {{1, 1, 1},  {1, 1, -1},  {1, -1, 1},  {1, -1, -1},
                                      {-1, 1, 1}, {-1, 1, -1}, {-1, -1, 1}, {-1, -1, -1}};
  // m = viewproj * transform, column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3 (glm_matmul)
  const v4f a0 = load4(viewproj), a1 = load4(viewproj + 4), a2 = load4(viewproj + 8), a3 = load4(viewproj + 12);
  v4f col[4];
  for (int j = 0; j < 4; j++) {
    const float* bj = obj.transform + 4 * j;
    v4f acc = a0 * splat4(bj[0]);
    acc = acc + a1 * splat4(bj[1]);
    acc = acc + a2 * splat4(bj[2]);
    acc = acc + a3 * splat4(bj[3]);
    col[j] = acc;
  }
  v4f mn = splat4(1.5f), mx = splat4(-1.5f);
  for (int c = 0; c < 8; c++) {
    float p[3];
    for (int k = 0; k < 3; k++) p[k] = obj.bounds.origin[k] + corners[c][k] * obj.bounds.extents[k];
    const v4f add0 = col[0] * splat4(p[0]) + col[1] * splat4(p[1]);
    const v4f add1 = col[2] * splat4(p[2]) + col[3] * splat4(1.0f);
    v4f v = add0 + add1;
    v = v / splat4(v[3]);  // lanes 0..2: x/w, y/w, z/w
    mn = select4(mn < v, mn, v);
    mx = select4(v < mx, mx, v);
  }
  return !(mn[2] > 1.f || mx[2] < 0.f || mn[0] > 1.f || mx[0] < -1.f || mn[1] > 1.f || mx[1] < -1.f);
}

}  // namespace svr
