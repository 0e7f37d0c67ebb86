"""The reference's own shader programs, EXECUTED, against the library's shader stages.

tests/golden/spv_exec.npz holds seeded inputs and the outputs the reference's committed SPIR-V modules
(/root/reference/shaders/*.spv) gave for them in the interpreter of tests/golden/spv_machine.py, in its two
arithmetic modes: `strict` (every SPIR-V operation rounds on its own) and `fused` (what a contracting compiler makes
of the same module; the modules carry no NoContraction).  Made by tests/golden/make_spv_exec.py inside the container;
nothing here reads /root/reference.

The checks feed the same inputs to a library exporting include/svr.h — the oracle here, the HIP library in
tests/test_spv_exec_gpu.py — through the per-stage hooks (svr_run_mesh_vert, svr_run_vertex_shader, the pixel trace)
and through whole passes whose fixed-function part is exact by construction (w = 1, constant varyings, one texel per
2x2 block), and require

  * bit-identical results to the `fused` execution (the arithmetic contract C0/C1/C10/C11/C14 of DESIGN.md IS that
    contraction pattern), signs of zero aside, and
  * agreement with the `strict` execution within the rounding of the operations that differ: a few ULP of the sum of
    magnitudes for the matrix chains (they cancel), the fragment colour's own roundings plus the dot product's (which cancels too), and the
    stored fp16 channel equal or adjacent.

What this pins: the programmable stages (rows a10, a13's shader half, a21-a23 of SURVEY 8a).  What it does not: the
fixed-function rules between them (snap, top-left, interpolation, LOD, filtering, blend rounding) stay contract-only.
"""
import os

import numpy as np
import pytest

import svr_testlib as T

FIX = os.path.join(T.GOLDEN_DIR, "spv_exec.npz")
EPS = 2.0 ** -24


@pytest.fixture(scope="module")
def fx():
    return dict(np.load(FIX))


def ordered(a):
    """float32 -> int64 that orders like the floats, +0 and -0 both 0"""
    b = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    return np.where(b < 0, -(b & 0x7fffffff), b)


def ulps(a, b):
    return np.abs(ordered(a) - ordered(b))


def same_bits(a, b):
    """bit-identical, except that +0 and -0 are one value"""
    return np.array_equal(ordered(a), ordered(b))


def f16_bits(a):
    with np.errstate(over="ignore"):
        return np.asarray(a, np.float32).astype(np.float16).view(np.uint16)


def f16_ordered(bits):
    b = np.asarray(bits, np.uint16).astype(np.int32)
    return np.where(b & 0x8000, -(b & 0x7fff), b)


def white_material(r, pkg, color_factors=(1, 1, 1, 1)):
    img = r.create_image(pkg.scenes.white_1x1(), mipmapped=False)
    smp = r.create_sampler(**pkg.scenes.SAMPLER_NEAREST)
    return r.write_material(pkg.abi.PASS_MAIN_COLOR, color_factors, img, smp)


def vertex_array(pkg, v12):
    v = np.zeros(len(v12), dtype=pkg.abi.VERTEX_DTYPE)
    v["position"], v["uv_x"], v["normal"], v["uv_y"], v["color"] = v12[:, 0:3], v12[:, 3], v12[:, 4:7], v12[:, 7], v12[:, 8:12]
    return v


def scene_of(pkg, viewproj=None, ambient=(0.1,) * 4, sun_dir=(0, 1, 0.5, 1), sun_col=(1,) * 4):
    eye = np.eye(4, dtype=np.float32).reshape(16)
    vp = eye if viewproj is None else viewproj
    return pkg.abi.scene_struct(eye, eye, vp, ambient, sun_dir, sun_col)


# ---------------------------------------------------------------- vertex programs
def check_mesh_vert(lib, pkg, fx):
    """shaders/mesh.vert:29-38 executed from mesh.vert.spv vs svr_run_mesh_vert"""
    worst = 0
    for k in range(len(fx["mesh_vert.viewproj"])):
        vp, world, cf = fx["mesh_vert.viewproj"][k], fx["mesh_vert.world"][k], fx["mesh_vert.color_factors"][k]
        v12 = fx["mesh_vert.vertices"][k]
        r = lib.create(16, 16)
        n = len(v12)
        mesh = r.upload_mesh(np.arange(n - n % 3, dtype=np.uint32), vertex_array(pkg, v12))
        mat = white_material(r, pkg, cf)
        clip, var = r.run_mesh_vert(mesh, 0, n, world, scene_of(pkg, vp), mat)
        r.close()
        assert same_bits(clip, fx["mesh_vert.clip.fused"][k]), f"case {k}: gl_Position differs from the fused execution"
        assert same_bits(var, fx["mesh_vert.varyings.fused"][k]), f"case {k}: varyings differ from the fused execution"
        # strict: the chains cancel, so the yardstick is the sum of magnitudes |VP| |W| |p| (8 roundings on the way)
        VP, W = np.abs(vp.astype(np.float64).reshape(4, 4).T), np.abs(world.astype(np.float64).reshape(4, 4).T)
        p = np.concatenate([np.abs(v12[:, 0:3].astype(np.float64)), np.ones((n, 1))], axis=1)
        mag = p @ (VP @ W).T
        err = np.abs(clip.astype(np.float64) - fx["mesh_vert.clip.strict"][k].astype(np.float64))
        assert np.all(err <= 8 * EPS * mag), f"case {k}: gl_Position vs strict: {np.max(err / mag) / EPS:.2f} eps of the magnitude"
        magn = np.abs(v12[:, 4:7].astype(np.float64)) @ W[:3, :3].T
        errn = np.abs(var[:, 0:3].astype(np.float64) - fx["mesh_vert.varyings.strict"][k][:, 0:3].astype(np.float64))
        assert np.all(errn <= 3 * EPS * magn), f"case {k}: normal vs strict"
        assert same_bits(var[:, 3:8], fx["mesh_vert.varyings.strict"][k][:, 3:8])  # one multiply / a copy: no freedom
        worst = max(worst, float(np.max(err / mag) / EPS))
    return worst


def check_tex_image_vert(lib, pkg, fx):
    """shaders/colored_triangle_mesh.vert:28-38 executed vs svr_run_vertex_shader"""
    for k in range(len(fx["tex_image_vert.render_matrix"])):
        m16, v12 = fx["tex_image_vert.render_matrix"][k], fx["tex_image_vert.vertices"][k]
        n = len(v12)
        r = lib.create(16, 16)
        mesh = r.upload_mesh(np.arange(n - n % 3, dtype=np.uint32), vertex_array(pkg, v12))
        clip, var = r.run_vertex_shader(pkg.abi.VS_COLORED_TRIANGLE_MESH, mesh, 0, n, m16)
        r.close()
        assert same_bits(clip, fx["tex_image_vert.clip.fused"][k])
        assert same_bits(var[:, 3:8], fx["tex_image_vert.varyings.fused"][k])
        assert same_bits(var[:, 3:8], fx["tex_image_vert.varyings.strict"][k])
        M = np.abs(m16.astype(np.float64).reshape(4, 4).T)
        p = np.concatenate([np.abs(v12[:, 0:3].astype(np.float64)), np.ones((n, 1))], axis=1)
        err = np.abs(clip.astype(np.float64) - fx["tex_image_vert.clip.strict"][k].astype(np.float64))
        assert np.all(err <= 4 * EPS * (p @ M.T))


def check_colored_triangle(lib, pkg, fx):
    """shaders/colored_triangle.vert:6-25 and .frag:9-12 executed vs the config-1 pipeline"""
    r = lib.create(256, 256)
    clip, var = r.run_vertex_shader(pkg.abi.VS_COLORED_TRIANGLE, 0, 0, 3, None)
    r.close()
    assert same_bits(clip, fx["colored_triangle.clip"])
    assert same_bits(var[:, 3:6], fx["colored_triangle.color"])
    assert np.all(var[:, 0:3] == 0) and np.all(var[:, 6:8] == 0)
    # the fragment program is vec4(inColor, 1)
    assert np.array_equal(fx["colored_triangle.frag_out"][:, :3], fx["colored_triangle.frag_in"]) and np.all(fx["colored_triangle.frag_out"][:, 3] == 1)
    # coverage of the pass from the EXECUTED vertices: exact integer edge functions on a 1/2-pixel grid, top-left rule
    size = 256
    out = T.render_config1(lib, size)
    c = fx["colored_triangle.clip"].astype(np.float64)
    xs, ys = (c[:, 0] / c[:, 3] + 1) * size / 2, (c[:, 1] / c[:, 3] + 1) * size / 2
    X, Y = np.rint(xs * 2).astype(np.int64), np.rint(ys * 2).astype(np.int64)   # half-pixel units: exact here
    area = (X[1] - X[0]) * (Y[2] - Y[0]) - (X[2] - X[0]) * (Y[1] - Y[0])
    if area < 0:
        X[[1, 2]], Y[[1, 2]] = X[[2, 1]], Y[[2, 1]]
    px, py = np.meshgrid(2 * np.arange(size) + 1, 2 * np.arange(size) + 1)
    inside = np.ones((size, size), bool)
    for i in range(3):
        a, b = (i + 1) % 3, (i + 2) % 3
        dx, dy = X[b] - X[a], Y[b] - Y[a]
        e = dx * (py - Y[a]) - dy * (px - X[a])
        top_left = (dy < 0) or (dy == 0 and dx > 0)
        inside &= (e >= 0) if top_left else (e > 0)
    rgba = out["rgba8"]
    covered = (rgba != 255).any(axis=2)
    assert np.array_equal(covered, inside)
    assert int(inside.sum()) == size * size // 2
    col = T.f16_bits_to_f32(out["color"])
    lam = np.zeros((size, size, 3))
    for i in range(3):
        a, b = (i + 1) % 3, (i + 2) % 3
        lam[..., i] = ((X[b] - X[a]) * (py - Y[a]) - (Y[b] - Y[a]) * (px - X[a])) / abs(area)
    expect = lam @ fx["colored_triangle.color"].astype(np.float64)   # barycentric mix of the executed colours
    assert np.allclose(col[inside][:, :3], expect[inside], atol=2e-3)
    assert np.all(col[inside][:, 3] == 1.0)


# ---------------------------------------------------------------- fragment programs through a pass
def block_quads(pkg, size, n, uv):
    """n quads, quad i covering exactly the 2x2 pixel block i of a size x size target at clip z = 0.5, w = 1, every
    corner carrying the same attributes: nothing between vertex program and fragment program rounds"""
    per_row = size // 2
    v = np.zeros((n, 4), dtype=pkg.abi.VERTEX_DTYPE)
    bx, by = np.arange(n) % per_row, np.arange(n) // per_row
    for c, (ox, oy) in enumerate(((0, 0), (2, 0), (0, 2), (2, 2))):
        v["position"][:, c, 0] = (2 * bx + ox) * (2.0 / size) - 1.0
        v["position"][:, c, 1] = (2 * by + oy) * (2.0 / size) - 1.0
        v["position"][:, c, 2] = 0.5
        v["uv_x"][:, c], v["uv_y"][:, c] = uv[:, 0], uv[:, 1]
    idx = (4 * np.arange(n)[:, None] + np.array([0, 1, 2, 2, 1, 3])[None, :]).astype(np.uint32)
    return v, idx.reshape(-1)


def texel_grid(texels):
    """n RGBA8 texels as a square power-of-two image and the uv of every texel's centre"""
    n = len(texels)
    side = 1 << int(np.ceil(np.log2(np.sqrt(n))))
    img = np.zeros((side, side, 4), np.uint8)
    img.reshape(-1, 4)[:n] = texels
    tx, ty = np.arange(n) % side, np.arange(n) // side
    uv = np.stack([(tx + 0.5) / side, (ty + 0.5) / side], axis=1).astype(np.float32)
    return img, uv


def check_mesh_frag(lib, pkg, fx, traced=12):
    """shaders/mesh.frag:12-19 executed from mesh.frag.spv vs a mesh pass (image) and the pixel trace (fp32)"""
    A = pkg.abi
    size = 64
    stats = {"pixels": 0, "f16_equal_strict": 0, "max_ulp_strict": 0}
    for k in range(len(fx["mesh_frag.normal"])):
        normal, color, texel = fx["mesh_frag.normal"][k], fx["mesh_frag.color"][k], fx["mesh_frag.texel"][k]
        n = len(normal)
        assert n == (size // 2) ** 2
        img, uv = texel_grid(texel)
        v, idx = block_quads(pkg, size, n, uv)
        v["normal"][:] = normal[:, None, :]
        v["color"][:, :, :3] = color[:, None, :]
        v["color"][:, :, 3] = 1.0
        r = lib.create(size, size)
        mesh = r.upload_mesh(idx, v.reshape(-1))
        image = r.create_image(img, mipmapped=False)
        smp = r.create_sampler(**pkg.scenes.SAMPLER_NEAREST)
        mat = r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), image, smp)
        obj = np.zeros(1, dtype=A.RENDER_OBJECT_DTYPE)
        obj["index_count"], obj["mesh"], obj["material"] = idx.size, mesh, mat
        obj["origin"], obj["extents"], obj["sphere_radius"] = (0, 0, 0.5), (1, 1, 0.5), 1.5
        obj["transform"] = np.eye(4, dtype=np.float32).reshape(16)
        scene = scene_of(pkg, None, fx["mesh_frag.ambient_color"][k], fx["mesh_frag.sunlight_direction"][k], fx["mesh_frag.sunlight_color"][k])
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, obj, None)
        r.sync()
        got = r.read_color()                                       # fp16 bit patterns [size, size, 4]
        blocks = got.reshape(size // 2, 2, size // 2, 2, 4).transpose(0, 2, 1, 3, 4).reshape(n, 4, 4)
        assert np.all(blocks == blocks[:, :1]), "a 2x2 block is not uniform"
        mine = blocks[:, 0]
        fused, strict = f16_bits(fx["mesh_frag.out.fused"][k]), f16_bits(fx["mesh_frag.out.strict"][k])
        bad = np.argwhere(mine != fused)
        assert bad.size == 0, f"case {k}: {len(bad)} stored channels differ from the fused execution, first {bad[:4].tolist()}"
        d = np.abs(f16_ordered(mine) - f16_ordered(strict))
        assert d.max() <= 1, f"case {k}: a stored channel is {d.max()} fp16 steps from the strict execution"
        stats["pixels"] += d.size
        stats["f16_equal_strict"] += int((d == 0).sum())
        # fp32, before the store: the traced invocation of a few blocks
        r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
        for i in np.linspace(0, n - 1, traced).astype(int):
            bx, by = i % (size // 2), i // (size // 2)
            r.trace_pixel(2 * bx + 1, 2 * by)
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, obj, None)
            r.sync()
            t = r.read_trace()
            tex = texel[i].astype(np.float32) * (np.float32(1) / np.float32(255))
            assert same_bits(t[11:14], tex[:3]), "the texel that reached the program is not the supplied one"  # (.xyz is all mesh.frag reads)
            assert same_bits(t[15:18], normal[i]), "the normal that reached the program is not the supplied one"
            assert same_bits(t[18:21], color[i] * tex[:3])
            assert same_bits(t[22:26], fx["mesh_frag.out.fused"][k][i])
            # strict: 2 ULP for the colour's own roundings + the dot product's, which cancels: 3 roundings of the sum of
            # its terms' magnitudes, relative to the light value it leaves
            want = fx["mesh_frag.out.strict"][k][i].astype(np.float64)
            L = fx["mesh_frag.sunlight_direction"][k][:3].astype(np.float64)
            terms = float(np.abs(normal[i].astype(np.float64) * L).sum())
            light = max(float(t[21]), 0.1)
            tol = (4 * EPS + 3 * EPS * terms / light) * np.abs(want)
            assert np.all(np.abs(t[22:26].astype(np.float64) - want) <= tol), f"case {k} block {i}: beyond the strict execution's rounding"
            stats["max_ulp_strict"] = max(stats["max_ulp_strict"], int(ulps(t[22:26], fx["mesh_frag.out.strict"][k][i]).max()))
        r.close()
    assert stats["f16_equal_strict"] >= 0.995 * stats["pixels"]
    return stats


def check_tex_image_frag(lib, pkg, fx):
    """shaders/tex_image.frag:10-12 executed vs the config-2 pipeline: the sampled RGBA, alpha included"""
    texel = fx["tex_image_frag.texel"]
    n, size = len(texel), 32
    assert n == (size // 2) ** 2
    img, uv = texel_grid(texel)
    v, idx = block_quads(pkg, size, n, uv)
    v["color"][:] = 1.0
    r = lib.create(size, size)
    mesh = r.upload_mesh(idx, v.reshape(-1))
    image = r.create_image(img, mipmapped=False)
    smp = r.create_sampler(**pkg.scenes.SAMPLER_NEAREST)
    r.clear_color((1, 1, 1, 1))
    r.draw_tex_image(mesh, 0, idx.size, np.eye(4, dtype=np.float32).reshape(16), image, smp)
    r.sync()
    got = r.read_color()
    r.close()
    blocks = got.reshape(size // 2, 2, size // 2, 2, 4).transpose(0, 2, 1, 3, 4).reshape(n, 4, 4)
    assert np.all(blocks == blocks[:, :1])
    assert np.array_equal(blocks[:, 0], f16_bits(fx["tex_image_frag.out"]))


# ---------------------------------------------------------------- compute programs
def check_gradient(lib, pkg, fx):
    """shaders/gradient_color.comp:14-27 executed vs svr_draw_background(GRADIENT)"""
    equal = total = 0
    for si, (w, h) in enumerate(fx["gradient.sizes"]):
        for di, data in enumerate(fx["gradient.data"]):
            r = lib.create(int(w), int(h))
            r.draw_background(pkg.abi.BACKGROUND_GRADIENT, data)
            r.sync()
            got = r.read_color()
            r.close()
            assert np.all(got == got[:, :1]), "the gradient depends on x"
            fused = f16_bits(fx["gradient.rows.%d.fused" % si][di])
            strict = f16_bits(fx["gradient.rows.%d.strict" % si][di])
            assert np.array_equal(got[:, 0], fused), f"size {w}x{h} data {di}: rows differ from the fused execution"
            d = np.abs(f16_ordered(got[:, 0]) - f16_ordered(strict))
            assert d.max() <= 1
            equal += int((d == 0).sum())
            total += d.size
    assert equal >= 0.995 * total
    return equal, total


def check_sky(lib, pkg, fx):
    """shaders/sky.comp executed (cos correctly rounded; Vulkan promises 2^-11 absolute) vs svr_draw_background(SKY):
    the star field is fract(415.9 (cos + cos)), so the last bit of a cosine moves a star's brightness in the fourth
    decimal and, rarely, a star across the threshold — compared with a tolerance, stars counted"""
    w, h = (int(x) for x in fx["sky.size"])
    r = lib.create(w, h)
    r.draw_background(pkg.abi.BACKGROUND_SKY, fx["sky.data"])
    r.sync()
    got = T.f16_bits_to_f32(r.read_color())
    r.close()
    want = fx["sky.image.strict"]
    close = np.isclose(got, want, atol=4e-3, rtol=2e-3).all(axis=2)
    assert close.mean() >= 0.99, f"{(~close).sum()} of {close.size} pixels differ"
    assert np.all(got[..., 3] == 1.0)
    gradient_only = want[..., 2] * 0 + (np.arange(h, dtype=np.float32)[:, None] * np.float32(0.4) / np.float32(h))
    stars_want = (want[..., 2] - gradient_only) > 1e-3
    stars_got = (got[..., 2] - gradient_only) > 1e-3
    assert (stars_want != stars_got).sum() <= max(2, 0.02 * stars_want.sum())
    return int(stars_want.sum()), float(close.mean())


# ---------------------------------------------------------------- the oracle against the executed programs
def test_fixture_is_complete(fx):
    for key in ("mesh_vert.clip.strict", "mesh_vert.clip.fused", "tex_image_vert.clip.fused", "colored_triangle.clip",
                "mesh_frag.out.strict", "mesh_frag.out.fused", "tex_image_frag.out", "gradient.rows.0.fused", "sky.image.strict"):
        assert key in fx and np.isfinite(fx[key]).all()
    # SURVEY 8c's shading constants come out of the executed mesh.frag: N = (0,1,0) -> 1.1, N = (1,0,0) -> 0.2
    for mode in ("strict", "fused"):
        out = fx["mesh_frag.out." + mode][0]
        assert np.allclose(out[0, :3], 1.1, atol=1e-6) and np.allclose(out[1, :3], 0.2, atol=1e-6) and out[0, 3] == 1.0
    # and the two modes really are two executions: they differ somewhere, by a few ULP (more where the dot product
    # of normal and sun direction cancels)
    d = ulps(fx["mesh_frag.out.strict"], fx["mesh_frag.out.fused"])
    assert 0 < d.max() <= 64 and np.median(d) <= 1


def test_mesh_vert_against_the_executed_module(oracle, pkg, fx):
    check_mesh_vert(oracle, pkg, fx)


def test_tex_image_vert_against_the_executed_module(oracle, pkg, fx):
    check_tex_image_vert(oracle, pkg, fx)


def test_colored_triangle_against_the_executed_modules(oracle, pkg, fx):
    check_colored_triangle(oracle, pkg, fx)


def test_mesh_frag_against_the_executed_module(oracle, pkg, fx):
    check_mesh_frag(oracle, pkg, fx)


def test_tex_image_frag_against_the_executed_module(oracle, pkg, fx):
    check_tex_image_frag(oracle, pkg, fx)


def test_gradient_against_the_executed_module(oracle, pkg, fx):
    check_gradient(oracle, pkg, fx)


def test_sky_against_the_executed_module(oracle, pkg, fx):
    check_sky(oracle, pkg, fx)
