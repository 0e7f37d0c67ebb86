"""fp32 restatement of the GLM calls the reference uses to produce GPUSceneData and node transforms.

Inputs of the draw path, not part of it: update_scene (src/vk_engine.cpp:1479-1512), Camera
(src/camera.cpp:54-66) and the loader's TRS composition (src/vk_loader.cpp:397-412).  glm is an
un-vendored, unpinned submodule of the reference, so the scalar operation order of
GLM 0.9.9 is restated from its published sources; tests pin it with the constants of SURVEY.md a14.
Matrices are numpy float32 arrays of shape (4,4) indexed m[col][row] like glm (flatten() gives the
column-major float[16] of the C ABI).
"""
import math

import numpy as np

f32 = np.float32


def identity():
    m = np.zeros((4, 4), dtype=f32)
    for i in range(4):
        m[i][i] = f32(1)
    return m


def radians(deg):
    return f32(deg) * f32(0.01745329251994329576923690768489)


def matmul(a, b):
    """glm::operator*(mat4, mat4): column j = ((A0*b0 + A1*b1) + A2*b2) + A3*b3."""
    out = np.empty((4, 4), dtype=f32)
    for j in range(4):
        acc = a[0] * b[j][0]
        acc = acc + a[1] * b[j][1]
        acc = acc + a[2] * b[j][2]
        acc = acc + a[3] * b[j][3]
        out[j] = acc
    return out


def matvec(m, v):
    """glm::operator*(mat4, vec4) = (m0*v0 + m1*v1) + (m2*v2 + m3*v3)."""
    v = np.asarray(v, dtype=f32)
    return (m[0] * v[0] + m[1] * v[1]) + (m[2] * v[2] + m[3] * v[3])


def perspective_rh_zo(fovy, aspect, z_near, z_far):
    fovy, aspect, z_near, z_far = f32(fovy), f32(aspect), f32(z_near), f32(z_far)
    tan_half = f32(math.tan(float(fovy / f32(2))))
    m = np.zeros((4, 4), dtype=f32)
    m[0][0] = f32(1) / (aspect * tan_half)
    m[1][1] = f32(1) / tan_half
    m[2][2] = z_far / (z_near - z_far)
    m[2][3] = -f32(1)
    m[3][2] = -(z_far * z_near) / (z_far - z_near)
    return m


def translate(m, v):
    v = np.asarray(v, dtype=f32)
    r = m.copy()
    r[3] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + m[3]
    return r


def scale(m, v):
    v = np.asarray(v, dtype=f32)
    r = np.empty((4, 4), dtype=f32)
    r[0] = m[0] * v[0]
    r[1] = m[1] * v[1]
    r[2] = m[2] * v[2]
    r[3] = m[3]
    return r


def rotate(m, angle, axis):
    a = f32(angle)
    c, s = f32(math.cos(float(a))), f32(math.sin(float(a)))
    axis = np.asarray(axis, dtype=f32)
    axis = axis * (f32(1) / f32(math.sqrt(float(np.dot(axis, axis)))))
    temp = (f32(1) - c) * axis
    R = np.zeros((4, 4), dtype=f32)
    R[0][0] = c + temp[0] * axis[0]
    R[0][1] = temp[0] * axis[1] + s * axis[2]
    R[0][2] = temp[0] * axis[2] - s * axis[1]
    R[1][0] = temp[1] * axis[0] - s * axis[2]
    R[1][1] = c + temp[1] * axis[1]
    R[1][2] = temp[1] * axis[2] + s * axis[0]
    R[2][0] = temp[2] * axis[0] + s * axis[1]
    R[2][1] = temp[2] * axis[1] - s * axis[0]
    R[2][2] = c + temp[2] * axis[2]
    r = np.empty((4, 4), dtype=f32)
    r[0] = m[0] * R[0][0] + m[1] * R[0][1] + m[2] * R[0][2]
    r[1] = m[0] * R[1][0] + m[1] * R[1][1] + m[2] * R[1][2]
    r[2] = m[0] * R[2][0] + m[1] * R[2][1] + m[2] * R[2][2]
    r[3] = m[3]
    return r


def angle_axis(angle, axis):
    """glm::angleAxis -> quaternion (w, x, y, z)."""
    a = f32(angle)
    s = f32(math.sin(float(a * f32(0.5))))
    c = f32(math.cos(float(a * f32(0.5))))
    axis = np.asarray(axis, dtype=f32)
    return np.array([c, axis[0] * s, axis[1] * s, axis[2] * s], dtype=f32)


def quat_to_mat4(q):
    w, x, y, z = [f32(t) for t in q]
    qxx, qyy, qzz = x * x, y * y, z * z
    qxz, qxy, qyz = x * z, x * y, y * z
    qwx, qwy, qwz = w * x, w * y, w * z
    one, two = f32(1), f32(2)
    m = identity()
    m[0][0] = one - two * (qyy + qzz)
    m[0][1] = two * (qxy + qwz)
    m[0][2] = two * (qxz - qwy)
    m[1][0] = two * (qxy - qwz)
    m[1][1] = one - two * (qxx + qzz)
    m[1][2] = two * (qyz + qwx)
    m[2][0] = two * (qxz + qwy)
    m[2][1] = two * (qyz - qwx)
    m[2][2] = one - two * (qxx + qyy)
    return m


def inverse(m):
    """glm::inverse(mat4) (cofactor expansion, detail::compute_inverse<4,4>)."""
    c00 = m[2][2] * m[3][3] - m[3][2] * m[2][3]
    c02 = m[1][2] * m[3][3] - m[3][2] * m[1][3]
    c03 = m[1][2] * m[2][3] - m[2][2] * m[1][3]
    c04 = m[2][1] * m[3][3] - m[3][1] * m[2][3]
    c06 = m[1][1] * m[3][3] - m[3][1] * m[1][3]
    c07 = m[1][1] * m[2][3] - m[2][1] * m[1][3]
    c08 = m[2][1] * m[3][2] - m[3][1] * m[2][2]
    c10 = m[1][1] * m[3][2] - m[3][1] * m[1][2]
    c11 = m[1][1] * m[2][2] - m[2][1] * m[1][2]
    c12 = m[2][0] * m[3][3] - m[3][0] * m[2][3]
    c14 = m[1][0] * m[3][3] - m[3][0] * m[1][3]
    c15 = m[1][0] * m[2][3] - m[2][0] * m[1][3]
    c16 = m[2][0] * m[3][2] - m[3][0] * m[2][2]
    c18 = m[1][0] * m[3][2] - m[3][0] * m[1][2]
    c19 = m[1][0] * m[2][2] - m[2][0] * m[1][2]
    c20 = m[2][0] * m[3][1] - m[3][0] * m[2][1]
    c22 = m[1][0] * m[3][1] - m[3][0] * m[1][1]
    c23 = m[1][0] * m[2][1] - m[2][0] * m[1][1]
    V = lambda *a: np.array(a, dtype=f32)
    fac0, fac1, fac2 = V(c00, c00, c02, c03), V(c04, c04, c06, c07), V(c08, c08, c10, c11)
    fac3, fac4, fac5 = V(c12, c12, c14, c15), V(c16, c16, c18, c19), V(c20, c20, c22, c23)
    vec0 = V(m[1][0], m[0][0], m[0][0], m[0][0])
    vec1 = V(m[1][1], m[0][1], m[0][1], m[0][1])
    vec2 = V(m[1][2], m[0][2], m[0][2], m[0][2])
    vec3 = V(m[1][3], m[0][3], m[0][3], m[0][3])
    inv0 = vec1 * fac0 - vec2 * fac1 + vec3 * fac2
    inv1 = vec0 * fac0 - vec2 * fac3 + vec3 * fac4
    inv2 = vec0 * fac1 - vec1 * fac3 + vec3 * fac5
    inv3 = vec0 * fac2 - vec1 * fac4 + vec2 * fac5
    sign_a, sign_b = V(1, -1, 1, -1), V(-1, 1, -1, 1)
    inv = np.stack([inv0 * sign_a, inv1 * sign_b, inv2 * sign_a, inv3 * sign_b]).astype(f32)
    row0 = V(inv[0][0], inv[1][0], inv[2][0], inv[3][0])
    dot0 = m[0] * row0
    dot1 = (dot0[0] + dot0[1]) + (dot0[2] + dot0[3])
    return (inv * (f32(1) / dot1)).astype(f32)


def camera_rotation(pitch, yaw):
    """Camera::get_rotation_matrix, src/camera.cpp:61-66."""
    pitch_q = angle_axis(pitch, (1, 0, 0))
    yaw_q = angle_axis(yaw, (0, -1, 0))
    return matmul(quat_to_mat4(yaw_q), quat_to_mat4(pitch_q))


def camera_view(position, pitch, yaw):
    """Camera::get_view_matrix, src/camera.cpp:54-59."""
    t = translate(identity(), position)
    return inverse(matmul(t, camera_rotation(pitch, yaw)))


def scene_data(view, window_w, window_h):
    """VulkanEngine::update_scene, src/vk_engine.cpp:1492-1498 -> (view, proj, viewproj, ambient,
    sunlight_direction, sunlight_color)."""
    proj = perspective_rh_zo(radians(70.0), f32(window_w) / f32(window_h), 10000.0, 0.1)
    proj[1][1] *= f32(-1)
    viewproj = matmul(proj, view)
    ambient = np.full(4, 0.1, dtype=f32)
    sun_color = np.full(4, 1.0, dtype=f32)
    sun_dir = np.array([0, 1, 0.5, 1.0], dtype=f32)
    return view, proj, viewproj, ambient, sun_dir, sun_color


def trs(translation, rotation_xyzw, scale_v):
    """Node local transform from glTF TRS, src/vk_loader.cpp:400-410: tm * rm * sm."""
    tm = translate(identity(), translation)
    x, y, z, w = rotation_xyzw
    rm = quat_to_mat4((w, x, y, z))
    sm = scale(identity(), scale_v)
    return matmul(matmul(tm, rm), sm)
