// svr_math.h — the handful of GLM calls the reference's host code makes around the draw path, restated
// in fp32 with GLM 0.9.9's scalar operation order (glm is an un-vendored, unpinned submodule of the
// reference).  Call sites restated: update_scene (src/vk_engine.cpp:1492-1495),
// Camera (src/camera.cpp:54-66), node TRS (src/vk_loader.cpp:397-412), MeshNode::Draw (:1717).
// Column-major like glm::mat4: m[c][r].
#pragma once
#include <cmath>
#include <cstring>

namespace svrm {

struct vec3 {
  float x = 0, y = 0, z = 0;
};
struct vec4 {
  float x = 0, y = 0, z = 0, w = 0;
};
struct quat {  // glm::quat(w, x, y, z)
  float w = 1, x = 0, y = 0, z = 0;
};
struct mat4 {
  float m[4][4];
  float* data() { return &m[0][0]; }
  const float* data() const { return &m[0][0]; }
};

inline mat4 identity() {
  mat4 r;
  std::memset(&r, 0, sizeof(r));
  r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0f;
  return r;
}
inline float radians(float deg) { return deg * 0.01745329251994329576923690768489f; }

// glm::operator*(mat4, mat4): column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3
inline mat4 mul(const mat4& a, const mat4& b) {
  mat4 r;
  for (int j = 0; j < 4; j++)
    for (int k = 0; k < 4; k++) {
      float acc = a.m[0][k] * b.m[j][0];
      acc = acc + a.m[1][k] * b.m[j][1];
      acc = acc + a.m[2][k] * b.m[j][2];
      acc = acc + a.m[3][k] * b.m[j][3];
      r.m[j][k] = acc;
    }
  return r;
}
// glm::operator*(mat4, vec4) = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)
inline vec4 mul(const mat4& m, const vec4& v) {
  float o[4];
  for (int k = 0; k < 4; k++) o[k] = (m.m[0][k] * v.x + m.m[1][k] * v.y) + (m.m[2][k] * v.z + m.m[3][k] * v.w);
  return vec4{o[0], o[1], o[2], o[3]};
}
// glm::perspectiveRH_ZO (GLM_FORCE_DEPTH_ZERO_TO_ONE, src/vk_engine.cpp:5)
inline mat4 perspective(float fovy, float aspect, float z_near, float z_far) {
  float tan_half = std::tan(fovy / 2.0f);
  mat4 r;
  std::memset(&r, 0, sizeof(r));
  r.m[0][0] = 1.0f / (aspect * tan_half);
  r.m[1][1] = 1.0f / tan_half;
  r.m[2][2] = z_far / (z_near - z_far);
  r.m[2][3] = -1.0f;
  r.m[3][2] = -(z_far * z_near) / (z_far - z_near);
  return r;
}
inline mat4 translate(const mat4& m, vec3 v) {
  mat4 r = m;
  for (int k = 0; k < 4; k++) r.m[3][k] = m.m[0][k] * v.x + m.m[1][k] * v.y + m.m[2][k] * v.z + m.m[3][k];
  return r;
}
inline mat4 scale(const mat4& m, vec3 v) {
  mat4 r;
  for (int k = 0; k < 4; k++) {
    r.m[0][k] = m.m[0][k] * v.x;
    r.m[1][k] = m.m[1][k] * v.y;
    r.m[2][k] = m.m[2][k] * v.z;
    r.m[3][k] = m.m[3][k];
  }
  return r;
}
inline quat angle_axis(float angle, vec3 axis) {
  float s = std::sin(angle * 0.5f);
  return quat{std::cos(angle * 0.5f), axis.x * s, axis.y * s, axis.z * s};
}
// glm::toMat4(quat)
inline mat4 to_mat4(quat q) {
  float qxx = q.x * q.x, qyy = q.y * q.y, qzz = q.z * q.z;
  float qxz = q.x * q.z, qxy = q.x * q.y, qyz = q.y * q.z;
  float qwx = q.w * q.x, qwy = q.w * q.y, qwz = q.w * q.z;
  mat4 r = identity();
  r.m[0][0] = 1.0f - 2.0f * (qyy + qzz);
  r.m[0][1] = 2.0f * (qxy + qwz);
  r.m[0][2] = 2.0f * (qxz - qwy);
  r.m[1][0] = 2.0f * (qxy - qwz);
  r.m[1][1] = 1.0f - 2.0f * (qxx + qzz);
  r.m[1][2] = 2.0f * (qyz + qwx);
  r.m[2][0] = 2.0f * (qxz + qwy);
  r.m[2][1] = 2.0f * (qyz - qwx);
  r.m[2][2] = 1.0f - 2.0f * (qxx + qyy);
  return r;
}
// glm::inverse(mat4): cofactor expansion (detail::compute_inverse<4,4>)
inline mat4 inverse(const mat4& mm) {
  const float(*m)[4] = mm.m;
  float c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3], c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3];
  float c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3], c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3];
  float c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3], c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3];
  float c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2], c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2];
  float c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2], c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3];
  float c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3], c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3];
  float c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2], c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2];
  float c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2], c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1];
  float c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1], c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1];
  float f0[4] = {c00, c00, c02, c03}, f1[4] = {c04, c04, c06, c07}, f2[4] = {c08, c08, c10, c11};
  float f3[4] = {c12, c12, c14, c15}, f4[4] = {c16, c16, c18, c19}, f5[4] = {c20, c20, c22, c23};
  float v0[4] = {m[1][0], m[0][0], m[0][0], m[0][0]}, v1[4] = {m[1][1], m[0][1], m[0][1], m[0][1]};
  float v2[4] = {m[1][2], m[0][2], m[0][2], m[0][2]}, v3[4] = {m[1][3], m[0][3], m[0][3], m[0][3]};
  const float sa[4] = {1, -1, 1, -1}, sb[4] = {-1, 1, -1, 1};
  mat4 inv;
  for (int k = 0; k < 4; k++) {
    float i0 = v1[k] * f0[k] - v2[k] * f1[k] + v3[k] * f2[k];
    float i1 = v0[k] * f0[k] - v2[k] * f3[k] + v3[k] * f4[k];
    float i2 = v0[k] * f1[k] - v1[k] * f3[k] + v3[k] * f5[k];
    float i3 = v0[k] * f2[k] - v1[k] * f4[k] + v2[k] * f5[k];
    inv.m[0][k] = i0 * sa[k];
    inv.m[1][k] = i1 * sb[k];
    inv.m[2][k] = i2 * sa[k];
    inv.m[3][k] = i3 * sb[k];
  }
  float d0 = m[0][0] * inv.m[0][0], d1 = m[0][1] * inv.m[1][0], d2 = m[0][2] * inv.m[2][0], d3 = m[0][3] * inv.m[3][0];
  float one_over_det = 1.0f / ((d0 + d1) + (d2 + d3));
  for (int c = 0; c < 4; c++)
    for (int k = 0; k < 4; k++) inv.m[c][k] = inv.m[c][k] * one_over_det;
  return inv;
}

}  // namespace svrm
