"""Known-answer tests that pin the CPU oracle (the reference ships no tests or goldens: SURVEY.md
§4, §8c, so the pins are analytic).  CPU only."""
import math

import numpy as np
import pytest

import __graft_entry__ as g
import scenarios as SC
import svr_testlib as T

pkg = g.load_package()
A, S, GL = pkg.abi, pkg.scenes, pkg.glmath


def rgba8(out):
    return out["rgba8"]


def f32c(out):
    return T.f16_bits_to_f32(out["color"])


# ---------------------------------------------------------------- a14: projection constants
def test_projection_constants():
    view = GL.camera_view((30.0, 0.0, -85.0), 0.0, 0.0)
    _, proj, viewproj, amb, sun_dir, sun_col = GL.scene_data(view, 1920, 1080)
    assert proj[0][0] == pytest.approx(0.803333223, rel=2e-7)
    assert proj[1][1] == pytest.approx(-1.42814803, rel=2e-7)
    assert proj[2][2] == pytest.approx(1.00000998e-5, rel=2e-7)
    assert proj[2][3] == -1.0
    assert proj[3][2] == pytest.approx(0.100000992, rel=2e-7)
    p1700 = GL.scene_data(view, 1700, 900)[1]
    assert p1700[0][0] == pytest.approx(0.756078362, rel=2e-7)
    p11 = GL.scene_data(view, 512, 512)[1]
    assert p11[0][0] == pytest.approx(1.42814803, rel=2e-7)
    assert list(amb) == [np.float32(0.1)] * 4 and list(sun_col) == [1, 1, 1, 1]
    assert list(sun_dir) == [0, 1, 0.5, 1]
    # view = inverse(T(pos) * R): with zero rotation it is a pure translation by -pos
    assert np.allclose(view[3][:3], [-30.0, 0.0, 85.0]) and np.allclose(view[:3, :3], np.eye(3))
    assert np.array_equal(viewproj, GL.matmul(proj, view))


def test_glm_inverse_and_rotation():
    m = GL.matmul(GL.translate(GL.identity(), (1, 2, 3)), GL.camera_rotation(0.3, -1.1))
    inv = GL.inverse(m)
    assert np.allclose(GL.matmul(m, inv), np.eye(4), atol=1e-6)
    # yaw rotates about (0,-1,0): yaw = +90 deg turns the camera's -z view direction to +x
    r = GL.camera_rotation(0.0, GL.radians(90.0))
    fwd = GL.matvec(r, (0, 0, -1, 0))
    assert np.allclose(fwd[:3], [1, 0, 0], atol=1e-6)


# ---------------------------------------------------------------- a12: depth table
@pytest.mark.parametrize("d", [1.0, 10.0, 85.0, 1000.0])
def test_depth_table(oracle, d):
    out = SC.depth_plane(oracle, d)
    z = out["depth"]
    expect = 0.1 * (1.0 / d - 1e-4) / (1.0 - 1e-5)
    assert np.all(z > 0)
    assert np.allclose(z, expect, rtol=2e-4)


# ---------------------------------------------------------------- a21 / config 1
def test_config1_colored_triangle(oracle):
    out = T.render_config1(oracle, 256, instrument=True)
    img = rgba8(out)
    covered = (img != 255).any(axis=2)
    assert int(covered.sum()) == 32768  # exactly half of 256x256
    assert out["stats"].rasterized_fragments == 32768 and out["stats"].triangle_count == 1
    # row j covers i in [128 - j/2, 127 + j/2] (even j) / [127 - m, 128 + m] (j = 2m+1): top-left rule
    for j in (0, 1, 2, 101, 254, 255):
        xs = np.nonzero(covered[j])[0]
        m = j // 2
        lo, hi = (128 - m, 127 + m) if j % 2 == 0 else (127 - m, 128 + m)
        if j == 0:
            assert xs.size == 0
        else:
            assert (xs.min(), xs.max(), xs.size) == (lo, hi, hi - lo + 1)
    # barycentric colour at pixel centres: vertices (256,256)=R, (0,256)=G, (128,0)=B
    c = f32c(out)
    for (i, j) in ((128, 128), (60, 200), (200, 250), (128, 3)):
        x, y = i + 0.5, j + 0.5
        lb = (256.0 - y) / 256.0
        lr = (x - 128.0 * lb) / 256.0
        lg = 1.0 - lb - lr
        assert np.allclose(c[j, i, :3], [lr, lg, lb], atol=2e-3)
        assert c[j, i, 3] == 1.0
    assert np.all(out["depth"] == 0.0)  # z = 0 passes GREATER_OR_EQUAL against the 0.0 clear
    assert np.all(c[~covered] == 1.0)   # colour loadOp LOAD keeps the white background


# ---------------------------------------------------------------- §8c shading constants
def test_shading_constants(oracle):
    up = SC.shading_constants(oracle, (0, 1, 0))
    # light = max(dot((0,1,0),(0,1,.5)), .1) = 1 -> 1*1*1 + 0.1 = 1.1 -> fp16 1.0996 -> UNORM8 255
    assert np.all(up["color"][..., :3] == np.float16(1.1).view(np.uint16))
    assert np.all(rgba8(up)[..., :3] == 255)
    side = SC.shading_constants(oracle, (1, 0, 0))
    # light = max(0, .1) = .1 -> .1 + .1 = .2 -> 51
    assert np.all(rgba8(side)[..., :3] == 51)
    assert np.all(rgba8(side)[..., 3] == 255)
    assert np.all(up["depth"] == 0.5)


# ---------------------------------------------------------------- fill rule
def test_shared_edge_hit_once(oracle):
    out = SC.shared_edge_additive(oracle)
    v = f32c(out)[..., 0]
    one = np.float32(np.float16(1.1))
    assert set(np.unique(v).tolist()) == {0.0, float(one)}  # never 2.2 (double hit)
    ys, xs = np.nonzero(v)
    # the covered set is exactly the rectangle of pixel centres inside the quad: no holes on the diagonal
    assert v[ys.min():ys.max() + 1, xs.min():xs.max() + 1].min() == one
    assert out["stats"].rasterized_fragments == int((v > 0).sum())


def test_fan_hit_once(oracle):
    out = SC.fan_additive(oracle)
    v = f32c(out)[..., 0]
    one = float(np.float32(np.float16(1.1)))
    assert set(np.unique(v).tolist()) <= {0.0, one}
    # interior of the ellipse is fully covered
    h, w = v.shape
    yy, xx = np.mgrid[0:h, 0:w]
    nx, ny = (xx + 0.5) / w * 2 - 1, (yy + 0.5) / h * 2 - 1
    inside = (nx / 0.93) ** 2 + (ny / 0.88) ** 2 < 0.9
    assert np.all(v[inside] == one)


# ---------------------------------------------------------------- depth test
def test_reversed_z_and_ties(oracle):
    red, green = [255, 0, 0], [0, 255, 0]
    a = rgba8(SC.depth_order(oracle, later_is_nearer=True))
    b = rgba8(SC.depth_order(oracle, later_is_nearer=False))
    # overlap region: pixel (16,16).  Larger depth = nearer wins regardless of order
    assert np.array_equal(a[16, 16, :3] > 0, np.array(green) > 0)
    assert np.array_equal(b[16, 16, :3] > 0, np.array(red) > 0)
    t = rgba8(SC.depth_tie(oracle))
    assert np.all((t[..., :3] > 0) == (np.array(green) > 0))  # equal depth: later draw wins


def test_transparent_pass(oracle):
    out = SC.transparent_layers(oracle)
    c = f32c(out)
    h16 = lambda x: float(np.float32(np.float16(x)))
    # sun (0,1,0), ambient 0, normal (0,1,0): shaded colour = vertex colour
    # right half, upper rows (y<16): background .25 + t1 (.3,0,0) [rows y<16 <-> clip y<0] + t2 (0,.2,0) + t3 where |x|<.5
    px = c[4, 30]  # x=30: outside t3, right of the opaque quad
    assert px[0] == pytest.approx(h16(h16(0.25 + 0.3)), abs=1e-3) and px[1] == pytest.approx(0.25 + 0.2, abs=2e-3)
    # left half: opaque .5 grey; t2 (z=.4) is behind it and fails the depth test; t1 (z=.6) passes
    px = c[4, 2]
    assert px[0] == pytest.approx(0.5 + 0.3, abs=2e-3) and px[1] == pytest.approx(0.5, abs=1e-3)
    # depth is written by the opaque quad only
    z = out["depth"]
    assert np.all(z[:, :16] == 0.5) and np.all(z[:, 16:] == 0.0)
    assert np.all(c[..., 3] == 1.0)


# ---------------------------------------------------------------- texturing
def test_sampler_behaviour(oracle):
    near = rgba8(SC.textured_plane(oracle, "nearest"))
    # engine default samplers have maxLod 0: base level only -> only the two texel colours appear
    cols = {tuple(x) for x in near.reshape(-1, 4)[:, :3].tolist()}
    assert cols <= {(0, 0, 0), (255, 0, 255)} and len(cols) == 2
    tri = rgba8(SC.textured_plane(oracle, "trilinear"))
    # 23.5 tiles of 32 texels over 64 pixels = 11.75 texels/pixel -> lod ~3.55 -> magenta/black averaged
    assert np.all(np.abs(tri[..., 0].astype(int) - 128) <= 2) and np.all(tri[..., 1] == 0)
    mag = rgba8(SC.textured_plane(oracle, "linear", tiles=0.11))
    assert len(np.unique(mag[..., 0])) > 8  # linear magnification interpolates between texels


def test_lod_polynomial(oracle):
    xs = np.concatenate([np.exp2(np.linspace(-20, 20, 4001)), [1.0, 2.0, 4.0, 0.5]]).astype(np.float32)
    got = np.array([oracle.lib.svr_oracle_lod(float(x)) for x in xs])
    assert np.max(np.abs(got - 0.5 * np.log2(xs.astype(np.float64)))) < 5e-5
    assert oracle.lib.svr_oracle_lod(1.0) == 0.0 and oracle.lib.svr_oracle_lod(4.0) == 1.0
    assert oracle.lib.svr_oracle_lod(0.0) == -50.0 and oracle.lib.svr_oracle_lod(float("inf")) == 50.0
    assert oracle.lib.svr_oracle_lod(float("nan")) == -50.0


def test_mip_chain(oracle):
    r = oracle.create(8, 8)
    img = r.create_image(S.checkerboard_32(), mipmapped=True)
    sizes = [r.read_image_level(img, l).shape[:2] for l in range(6)]
    assert sizes == [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2), (1, 1)]  # floor(log2(32))+1 levels
    l1 = r.read_image_level(img, 1)
    # every 2x2 block holds two magenta and two black texels: (255+255)/4 = 127.5 -> ties-to-even 128
    assert np.all(l1[..., 0] == 128) and np.all(l1[..., 1] == 0) and np.all(l1[..., 2] == 128) and np.all(l1[..., 3] == 255)
    with pytest.raises(pkg.SvrError):
        r.read_image_level(img, 6)
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (8, 16, 4), dtype=np.uint8)   # non-square: 16x8
    im2 = r.create_image(tex, mipmapped=True)
    l1 = r.read_image_level(im2, 1)
    ref = tex.astype(np.int64).reshape(4, 2, 8, 2, 4).sum(axis=(1, 3))
    q, rem = ref // 4, ref % 4
    ref8 = q + ((rem == 3) | ((rem == 2) & (q % 2 == 1)))
    assert np.array_equal(l1, ref8.astype(np.uint8))
    assert [r.read_image_level(im2, l).shape[:2] for l in range(5)] == [(8, 16), (4, 8), (2, 4), (1, 2), (1, 1)]
    r.close()


def test_fp16_conversion(oracle):
    rng = np.random.default_rng(5)
    bits = rng.integers(0, 2 ** 32, 20000, dtype=np.uint64).astype(np.uint32)
    h = np.arange(0, 0x7c00, 7, dtype=np.uint16)
    ties = (h.view(np.float16).astype(np.float32).view(np.uint32).astype(np.uint64) + 0x1000).astype(np.uint32)
    vals = np.concatenate([bits, ties, ties - 1, ties + 1]).view(np.float32)
    vals = vals[np.isfinite(vals)]
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    got = np.array([oracle.lib.svr_oracle_f32_to_f16(float(v)) for v in vals], dtype=np.uint16)
    assert np.array_equal(got, ref)
    back = np.array([oracle.lib.svr_oracle_f16_to_f32(int(x)) for x in h], dtype=np.float32)
    assert np.array_equal(back, h.view(np.float16).astype(np.float32))


# ---------------------------------------------------------------- clipping
def test_floor_clipped_both_ends(oracle):
    out = SC.perspective_floor(oracle)
    z = out["depth"]
    h, w = z.shape
    covered = z > 0
    # camera 1.5 above an (effectively) infinite plane, horizontal view: everything below the
    # horizon row is floor; the far plane (10000) cuts a sub-pixel sliver at the horizon
    assert not covered[: h // 2 - 1].any()
    assert covered[h // 2 + 1:].all()
    # depth grows towards the bottom of the screen (nearer = larger, reversed-Z)
    col = z[h // 2 + 1:, w // 2]
    assert np.all(np.diff(col) > 0)
    # bottom row: ray through pixel centre hits the floor at distance d along -z_view
    py = h - 0.5
    ndc_y = py / h * 2 - 1
    tan_half = math.tan(math.radians(35.0))
    d = 1.5 / (ndc_y * tan_half)
    assert col[-1] == pytest.approx(0.1 * (1.0 / d - 1e-4) / (1 - 1e-5), rel=1e-3)


def test_near_clip_wall_is_watertight(oracle):
    out = SC.near_clip_wall(oracle)
    z = out["depth"]
    assert (z > 0).mean() > 0.5
    assert z.max() <= 1.0
    # the clipped polygon is a fan of triangles: every covered pixel hit exactly once
    assert out["stats"].rasterized_fragments == int((z > 0).sum())


# ---------------------------------------------------------------- is_visible (a2)
def vis(oracle, origin, extents, viewproj):
    ro = SC.render_object(1, 1, 0, 3, origin=origin, extents=extents)
    import ctypes as C
    vp = (C.c_float * 16)(*np.asarray(viewproj, dtype=np.float32).reshape(16).tolist())
    return bool(oracle.lib.svr_oracle_is_visible(ro.ctypes.data, vp))


def test_is_visible(oracle):
    view = GL.camera_view((0.0, 0.0, 0.0), 0.0, 0.0)
    vp = GL.scene_data(view, 1920, 1080)[2]
    assert vis(oracle, (0, 0, -10), (1, 1, 1), vp)            # in front
    assert not vis(oracle, (100, 0, -10), (1, 1, 1), vp)      # far to the right: min.x > 1
    assert not vis(oracle, (0, -100, -10), (1, 1, 1), vp)     # far below
    assert vis(oracle, (0, 0, -10), (1000, 1000, 1), vp)      # huge box around the view axis
    # quirk: min/max start at +-1.5, so a box whose corners all project beyond |1.5| in x on one
    # side leaves min.x = 1.5 > 1 -> culled; on the other side max.x stays -1.5 < -1 -> culled
    assert not vis(oracle, (40, 0, -10), (1, 1, 1), vp)
    assert not vis(oracle, (-40, 0, -10), (1, 1, 1), vp)
    # quirk: no w<=0 guard.  A box fully behind the camera divides by negative w; its z/w > 1 here
    assert not vis(oracle, (0, 0, 10), (1, 1, 1), vp)


def test_cull_and_sort_stats(oracle):
    out = T.render_sponza(oracle, 96, 54, lod=8, tex_size=16, instrument=True)
    st = out["stats"]
    assert st.culled_draws > 0  # the part of the atrium behind the camera
    assert st.drawcall_count == out["n_opaque"] - st.culled_draws + out["n_transparent"]  # transparent never culled
    assert st.triangle_count > 0 and st.binned_triangles > 0


# ---------------------------------------------------------------- ABI behaviour
def test_scissor_band_matches_full_frame(oracle):
    full = SC.random_soup(oracle, seed=5)
    band = SC.random_soup(oracle, seed=5, scissor=(13, 21, 101, 37))
    ys, xs = slice(21, 58), slice(13, 114)
    assert np.array_equal(full["color"][ys, xs], band["color"][ys, xs])
    assert np.array_equal(full["depth"][ys, xs], band["depth"][ys, xs])
    mask = np.ones(full["depth"].shape, bool)
    mask[ys, xs] = False
    assert np.all(band["depth"][mask] == 0.0)                                 # untouched: context starts zeroed
    # svr_clear_color fills the rows of the scissor: clear colour beside the rectangle, zeros above/below
    one = np.float16(1.0).view(np.uint16)
    assert np.all(band["color"][ys, :13] == one) and np.all(band["color"][ys, 114:] == one)
    assert np.all(band["color"][:21] == 0) and np.all(band["color"][58:] == 0)


def test_threads_do_not_change_the_image(oracle):
    a = T.render_sponza(oracle, 128, 72, lod=8, tex_size=32)
    b = T.render_sponza(oracle, 128, 72, lod=8, tex_size=32, threads=5)
    T.assert_images_identical(a["color"], b["color"], "threads colour")
    T.assert_images_identical(a["depth"], b["depth"], "threads depth")
    assert a["stats"].rasterized_fragments == b["stats"].rasterized_fragments


def test_ragged_and_empty(oracle):
    out = SC.ragged_draws(oracle)
    assert out["stats"].triangle_count == 1 + 0 + 2 + 0 and out["stats"].drawcall_count == 4
    img = rgba8(out)
    # red quad = screen [0,24)^2; index_count 5 -> only its first triangle (the half above the diagonal)
    assert img[4, 20, 0] > 0 and img[4, 20, 2] == 0
    assert tuple(img[20, 4]) == (255, 255, 255, 255)
    assert img[30, 40, 2] > 0 and img[40, 30, 2] > 0 and img[30, 40, 0] == 0   # blue quad, both triangles
    e = SC.empty_frame(oracle)
    assert np.all(e["depth"] == 0) and np.all(rgba8(e) == 255)


def test_errors(oracle):
    r = oracle.create(16, 16)
    with pytest.raises(pkg.SvrError) as ei:
        r.upload_mesh(np.array([0, 1, 5], np.uint32), SC.clip_quad(-1, -1, 1, 1, 0.5))
    assert ei.value.code == -1
    mesh = r.upload_mesh(SC.QUAD_IDX, SC.clip_quad(-1, -1, 1, 1, 0.5))
    img = r.create_image(S.white_1x1())
    smp = r.create_sampler()
    mo = r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), img, smp)
    mt = r.write_material(A.PASS_TRANSPARENT, (1, 1, 1, 1), img, smp)
    sc = SC.identity_scene()
    for bad, code in ((SC.render_object(7, mo, 0, 6), -4), (SC.render_object(mesh, 9, 0, 6), -4),
                      (SC.render_object(mesh, mo, 3, 6), -1), (SC.render_object(mesh, mt, 0, 6), -1)):
        with pytest.raises(pkg.SvrError) as ei:
            r.draw_geometry(sc, SC.objs([bad]))
        assert ei.value.code == code
    with pytest.raises(pkg.SvrError):
        r.draw_geometry(sc, None, SC.objs([SC.render_object(mesh, mo, 0, 6)]))  # opaque material in the transparent list
    with pytest.raises(pkg.SvrError):
        r.set_scissor(8, 8, 16, 4)
    with pytest.raises(pkg.SvrError):
        r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), 99, smp)
    with pytest.raises(pkg.SvrError):
        oracle.create(0, 16)
    r.close()


def test_rgba8_target(oracle):
    a = SC.random_soup(oracle, seed=11)
    b = SC.random_soup(oracle, seed=11, color_format=A.COLOR_RGBA8)
    assert np.array_equal(a["depth"], b["depth"])
    # opaque-only pixels: RGBA8 target == fp16 target read back as RGBA8 up to the double rounding (1 LSB)
    d = np.abs(a["rgba8"].astype(int) - b["rgba8"].astype(int))
    assert np.percentile(d, 99) <= 1


# ---------------------------------------------------------------- a23 / §8f-2: draw_background, copy_image
def test_background_gradient_values(oracle):
    """gradient_color.comp: mix(data1, data2, float(y)/height); the engine default (both white) is the
    clear; a red->blue gradient has analytic rows."""
    r = oracle.create(8, 4)
    r.draw_background(A.BACKGROUND_GRADIENT, A.GRADIENT_DEFAULT)
    assert np.all(r.read_color(as_rgba8=True) == 255)
    data = (1.0, 0.0, 0.0, 1.0, 0.0, 0.0, 1.0, 1.0) + (0.0,) * 8
    r.draw_background(A.BACKGROUND_GRADIENT, data)
    c = T.f16_bits_to_f32(r.read_color())
    for y in range(4):
        blend = np.float32(y) / np.float32(4)
        assert np.all(c[y, :, 0] == np.float16(np.float32(1) - blend)) and np.all(c[y, :, 2] == np.float16(blend))
        assert np.all(c[y, :, 1] == 0) and np.all(c[y, :, 3] == 1)
    r.close()


def test_background_respects_the_scissor_rows(oracle):
    r = oracle.create(16, 16)
    r.clear_color((0.0, 0.0, 0.0, 1.0))
    r.set_scissor(0, 4, 16, 8)
    r.draw_background(A.BACKGROUND_GRADIENT, A.GRADIENT_DEFAULT)
    c = r.read_color(as_rgba8=True)
    assert np.all(c[4:12, :, :3] == 255) and np.all(c[:4, :, :3] == 0) and np.all(c[12:, :, :3] == 0)
    r.close()


def test_background_sky_properties(oracle):
    """sky.comp: colour = data1.xyz * y / height + stars, alpha 1; stars are sparse at threshold 0.97 and
    the field does not depend on how the frame is cut into scissor bands."""
    W, H = 96, 64
    r = oracle.create(W, H)
    r.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
    full = T.f16_bits_to_f32(r.read_color())
    assert np.all(full[..., 3] == 1.0)
    base = (np.float32(0.4) * np.arange(H, dtype=np.float32)) / np.float32(H)
    star = full[..., 2] - base[:, None].astype(np.float16).astype(np.float32)
    assert (star > 0.01).mean() < 0.1 and (star > 0.01).sum() > 0       # sparse, but there
    assert np.all(full[..., 2] >= base[:, None].astype(np.float16).astype(np.float32) - 1e-3)
    r.clear_color((0, 0, 0, 0))
    for y0 in range(0, H, 16):
        r.set_scissor(0, y0, W, 16)
        r.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
    assert np.array_equal(T.f16_bits_to_f32(r.read_color()), full)
    r.close()


def test_swapchain_blit(oracle):
    """copy_image: identity extent = the plain format conversion (and B8G8R8A8 is its channel swap);
    a 2:1 reduction of a 2x2 checker averages to the midpoint; magnification interpolates."""
    out = T.render_config1(oracle, 64)
    r = oracle.create(64, 64)
    r.clear_color((1, 1, 1, 1))
    r.draw_colored_triangle()
    rgba = r.read_swapchain(64, 64, A.SWAPCHAIN_R8G8B8A8)
    bgra = r.read_swapchain(64, 64, A.SWAPCHAIN_B8G8R8A8)
    assert np.array_equal(rgba, out["rgba8"])
    assert np.array_equal(bgra[..., [2, 1, 0, 3]], rgba)
    r.close()
    r = oracle.create(4, 4)
    for y in range(4):     # rows alternate black / white through one-row scissors
        r.set_scissor(0, y, 4, 1)
        r.clear_color((1, 1, 1, 1) if y & 1 else (0, 0, 0, 1))
    half = r.read_swapchain(4, 2, A.SWAPCHAIN_R8G8B8A8)
    assert np.all(half[..., :3] == 128) and np.all(half[..., 3] == 255)   # 0.5 * 255 = 127.5 -> RNE 128
    up = r.read_swapchain(4, 8, A.SWAPCHAIN_R8G8B8A8)[:, 0, 0].astype(int)
    assert up[0] == 0 and up[-1] == 255 and np.all(np.diff(up[:3]) >= 0)
    with pytest.raises(A.SvrError):
        r.read_swapchain(0, 4)
    r.close()
