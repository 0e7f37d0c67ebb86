"""Parity tests proper: the HIP path (through the C ABI of libsvr_hip.so) against the CPU oracle on
the same inputs, bit for bit (colour fp16/unorm8 bits, depth f32 bits, mip texels, vertex-stage
floats), plus size-independent properties at the BASELINE sizes.  Needs a real MI355X."""
import numpy as np
import pytest

import __graft_entry__ as g
import scenarios as SC
import svr_testlib as T

pkg = g.load_package()
A, S, GL = pkg.abi, pkg.scenes, pkg.glmath
pytestmark = pytest.mark.gpu


def both(fn, hip, oracle, *a, **k):
    return fn(hip, *a, **k), fn(oracle, *a, **k)


def assert_same(a, b, what, stats=True):
    T.assert_images_identical(a["color"], b["color"], what + " colour")
    T.assert_images_identical(a["depth"], b["depth"], what + " depth")
    T.assert_images_identical(a["rgba8"], b["rgba8"], what + " rgba8 readback")
    if stats:
        sa, sb = a["stats"], b["stats"]
        assert (sa.triangle_count, sa.drawcall_count, sa.culled_draws) == (sb.triangle_count, sb.drawcall_count, sb.culled_draws)
        assert sa.rasterized_fragments == sb.rasterized_fragments, what + " rasterized fragments"
        assert sa.binned_triangles == sb.binned_triangles, what + " binned triangles"


def test_backend_is_the_hip_library(hip):
    assert hip.backend == "hip-gfx950"
    assert hip.path.endswith("libsvr_hip.so")


@pytest.mark.parametrize("name", sorted(SC.SCENARIOS))
def test_scenario(hip, oracle, name):
    a, b = both(SC.SCENARIOS[name], hip, oracle)
    assert_same(a, b, name)


def test_config1_colored_triangle(hip, oracle):
    a, b = both(T.render_config1, hip, oracle, 256, instrument=True)
    assert_same(a, b, "config1")
    assert a["stats"].rasterized_fragments == 32768


def test_config2_textured_cube_full_size(hip, oracle):
    a, b = both(T.render_config2, hip, oracle, 1920, 1080, instrument=True)
    assert_same(a, b, "config2")
    assert a["stats"].triangle_count == 12


@pytest.mark.parametrize("size,lod,tex", [((256, 144), 8, 64), ((640, 360), 4, 128), ((333, 187), 2, 64)])
def test_config3_sponza_reduced(hip, oracle, size, lod, tex):
    a, b = both(T.render_sponza, hip, oracle, size[0], size[1], lod=lod, tex_size=tex, instrument=True)
    assert_same(a, b, f"config3 {size} lod {lod}")


def test_config3_sponza_full_geometry_1080p(hip, oracle):
    """All 262,144 triangles at 1920x1080 (the oracle runs row-band parallel to stay in seconds)."""
    a = T.render_sponza(hip, 1920, 1080, lod=1, tex_size=256, instrument=True)
    b = T.render_sponza(oracle, 1920, 1080, lod=1, tex_size=256, instrument=True, threads=16)
    assert_same(a, b, "config3 1080p")
    assert a["stats"].triangle_count > 200000


def test_queue_overflow_replays_are_invisible(hip, oracle):
    """Internal queues (clip queue, clipper records, pair/bin lists) sized far too small: the passes
    overflow, write nothing, and are replayed with grown queues; the frame is the oracle's."""
    cam = ((2.5, 1.0, -5.5), 0.2, 1.0)  # inside a column: plenty of clipped triangles
    b = T.render_sponza(oracle, 320, 180, lod=4, tex_size=64, camera=cam, instrument=True)
    for caps in (1, 64, 3000):
        a = T.render_sponza(hip, 320, 180, lod=4, tex_size=64, camera=cam, instrument=True, queue_caps=caps)
        assert_same(a, b, f"queue caps {caps}")


def _unfenced_sequence(lib, caps):
    cam1, cam2 = ((2.5, 1.0, -5.5), 0.2, 1.0), ((30.0, 8.0, 9.7), -0.3, 3.0)
    r, scene1, opaque, transparent = T.setup_sponza(lib, 320, 180, lod=4, tex_size=64, camera=cam1)
    scene2 = S.scene_data_struct(*cam2, 320, 180)
    if caps is not None:
        r.set_option(A.OPT_QUEUE_CAPS, caps)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene1, opaque, transparent)
    r.set_scissor(0, 0, 160, 90)
    r.clear_color((0.25, 0.5, 0.75, 1.0))
    r.draw_colored_triangle()
    r.set_scissor(40, 60, 200, 100)
    r.draw_geometry(scene2, opaque, transparent)
    out = T._finish(r)
    r.close()
    return out


def test_overflow_inside_an_unfenced_sequence(hip, oracle):
    """Five target-writing operations enqueued without a fence, the first pass overflowing its queues:
    the later clears and passes must not land on top of a frame that misses it (operation log)."""
    b = _unfenced_sequence(oracle, None)
    for caps in (64, 2000):
        a = _unfenced_sequence(hip, caps)
        assert_same(a, b, f"unfenced sequence, caps {caps}", stats=False)
        assert a["stats"].replayed_passes >= 1
    a = _unfenced_sequence(hip, None)
    assert_same(a, b, "unfenced sequence, default caps", stats=False)
    assert a["stats"].replayed_passes == 0


def _background_and_blit(lib, w, h, fmt, effect, data, blits):
    r = lib.create(w, h, fmt)
    r.clear_color((0.2, 0.4, 0.6, 1.0))
    r.set_scissor(0, 3, w, h - 7)
    r.draw_background(effect, data)
    r.set_scissor(0, 0, w, h)
    r.draw_colored_triangle()
    out = {"color": r.read_color()}
    for k, (dw, dh, f) in enumerate(blits):
        out[f"blit{k}"] = r.read_swapchain(dw, dh, f)
    r.close()
    return out


@pytest.mark.parametrize("fmt", [0, 1])
@pytest.mark.parametrize("effect", ["gradient", "sky"])
def test_background_and_swapchain_blit(hip, oracle, fmt, effect):
    """draw_background's two effects and the scaling copy_image, bit for bit (odd sizes, both target
    formats, both swapchain channel orders, minification and magnification)."""
    data = (0.9, 0.1, 0.3, 1.0, 0.05, 0.6, 1.7, 0.5) + (0.0,) * 8 if effect == "gradient" else A.SKY_DEFAULT
    eff = A.BACKGROUND_GRADIENT if effect == "gradient" else A.BACKGROUND_SKY
    blits = [(333, 187, 0), (333, 187, 1), (160, 90, 0), (500, 401, 1), (1, 1, 0), (1024, 7, 0)]
    a, b = both(_background_and_blit, hip, oracle, 333, 187, fmt, eff, data, blits)
    for k in a:
        T.assert_images_identical(a[k], b[k], f"{effect} fmt {fmt} {k}")


def test_sky_at_4k_matches_the_oracle_rows(hip, oracle):
    """The star hash amplifies any difference in cos(): compare a band of rows at the far end of a
    3840x2160 target (largest arguments) against the oracle."""
    outs = []
    for lib in (hip, oracle):
        r = lib.create(3840, 2160)
        r.set_scissor(0, 2100, 3840, 60)
        r.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
        outs.append(r.read_color()[2100:])
        r.close()
    T.assert_images_identical(outs[0], outs[1], "sky rows 2100..2159 at 4K")


def test_copy_to_swapchain_device_image(hip):
    """svr_copy_to_swapchain writes caller-owned device memory in stream order."""
    import torch
    r = hip.create(256, 256)
    r.clear_color((1, 1, 1, 1))
    r.draw_colored_triangle()
    dst = torch.zeros((128, 128, 4), dtype=torch.uint8, device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    r.copy_to_swapchain(dst.data_ptr(), 128, 128, A.SWAPCHAIN_B8G8R8A8)
    r.sync()
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy(), r.read_swapchain(128, 128, A.SWAPCHAIN_B8G8R8A8))
    r.close()


def test_identity_blit_writes_the_scissor_rows(hip):
    """The multi-GPU present step: an identity-sized svr_copy_to_swapchain writes exactly the rows of
    the scissor, so bands presented one by one assemble the whole swapchain image."""
    import torch
    r = hip.create(320, 200)
    r.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
    r.draw_colored_triangle()
    whole = r.read_swapchain(320, 200, A.SWAPCHAIN_B8G8R8A8)
    dst = torch.full((200, 320, 4), 7, dtype=torch.uint8, device="cuda")
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    for y0, n in ((0, 67), (67, 67), (134, 66)):
        r.set_scissor(0, y0, 320, n)
        r.copy_to_swapchain(dst.data_ptr(), 320, 200, A.SWAPCHAIN_B8G8R8A8)
        r.sync()
        got = dst.cpu().numpy()
        assert np.array_equal(got[:y0 + n], whole[:y0 + n]) and np.all(got[y0 + n:] == 7)
    r.close()


def test_device_flatten_matches_the_host_path(hip, oracle):
    """SVR_OPT_DEVICE_FLATTEN = 1: is_visible, the sort and the draw records on the device (k_flatten.hip).
    Frames, fragment counts and the three late stats equal the oracle's (whose host loop is the
    reference's), for the plain scene, the 16-instance grid (5408 objects: above the automatic
    threshold too), cameras inside geometry, and with queue overflows replayed."""
    cases = [dict(lod=4, tex_size=64), dict(lod=8, tex_size=32, camera=S.config5_camera(), instances=S.config5_instances()),
             dict(lod=4, tex_size=64, camera=((2.5, 1.0, -5.5), 0.2, 1.0)), dict(lod=4, tex_size=64, camera=((54.5, 15.5, 0.0), -1.2, 4.6))]
    for k, kw in enumerate(cases):
        b = T.render_sponza(oracle, 320, 180, instrument=True, threads=8, **kw)
        for mode in (1, 0):
            a = T.render_sponza(hip, 320, 180, instrument=True, device_flatten=mode, **kw)
            assert_same(a, b, f"device flatten case {k} mode {mode}")
    b = T.render_sponza(oracle, 320, 180, lod=4, tex_size=64, instrument=True)
    a = T.render_sponza(hip, 320, 180, lod=4, tex_size=64, instrument=True, device_flatten=1, queue_caps=64)
    assert_same(a, b, "device flatten with replays")
    assert a["stats"].replayed_passes >= 1


def _all_culled_after_a_full_pass(lib):
    away = ((-200.0, 2.0, 0.0), 0.0, float(GL.radians(-90.0)))  # behind the atrium, looking further away
    r, scene, opaque, transparent = T.setup_sponza(lib, 320, 180, lod=4, tex_size=64)
    r.set_option(A.OPT_DEVICE_FLATTEN, 1)
    r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)      # leaves draw records and chunks of 338 objects behind
    r.draw_geometry(scene, opaque, transparent)      # ... in both sets of per-pass buffers
    r.clear_color((0.5, 0.25, 0.125, 1.0))
    r.draw_geometry(S.scene_data_struct(*away, 320, 180), opaque[:77], None)  # every object culled, no transparent list
    out = T._finish(r)
    r.close()
    return out


def test_device_flatten_with_every_object_culled(hip, oracle):
    """A device-flattened pass whose objects are all rejected by is_visible writes no chunk at all; the
    setup kernel's grid (sized by the host's upper bound) must not follow stale chunk records."""
    a, b = both(_all_culled_after_a_full_pass, hip, oracle)
    assert_same(a, b, "all culled, device flatten")
    assert a["stats"].culled_draws == 77 and a["stats"].drawcall_count == 0
    assert np.all(a["depth"] == 0)


def test_random_cameras(hip, oracle):
    """Seeded camera fuzz over the atrium (positions inside and outside the geometry, any pitch/yaw,
    odd extents): the frame is the oracle's, bit for bit, every time."""
    rng = np.random.default_rng(20261004)
    sizes = [(256, 144), (203, 117), (320, 96)]
    for k in range(18):
        pos = (float(rng.uniform(-5, 65)), float(rng.uniform(0.2, 17)), float(rng.uniform(-14, 14)))
        cam = (pos, float(rng.uniform(-1.4, 1.4)), float(rng.uniform(0, 6.283)))
        w, h = sizes[k % 3]
        a, b = both(T.render_sponza, hip, oracle, w, h, lod=8 if k % 2 else 4, tex_size=64, camera=cam, instrument=True)
        assert_same(a, b, f"fuzz camera {k} {cam} {w}x{h}")


def test_rgba8_target(hip, oracle):
    a, b = both(T.render_sponza, hip, oracle, 320, 180, lod=8, tex_size=64, color_format=A.COLOR_RGBA8, instrument=True)
    assert_same(a, b, "config3 rgba8")


def test_scissor_bands_tile_the_frame(hip, oracle):
    """The multi-GPU decomposition: every band rendered with its own scissor equals that band of the
    full frame (bands start at rows that are not multiples of the 32-pixel tile)."""
    full = T.render_sponza(oracle, 480, 270, lod=8, tex_size=64)
    for k in range(5):
        y0, y1 = k * 54, (k + 1) * 54
        band = T.render_sponza(hip, 480, 270, lod=8, tex_size=64, scissor=(0, y0, 480, 54))
        assert np.array_equal(band["color"][y0:y1], full["color"][y0:y1]), f"band {k} colour"
        assert np.array_equal(band["depth"][y0:y1], full["depth"][y0:y1]), f"band {k} depth"
        assert np.all(band["depth"][:y0] == 0) and np.all(band["depth"][y1:] == 0)


def test_quarters_with_clipped_transparent_triangles(hip, oracle):
    """Cameras inside the curtains' planes, looking along them: the near plane cuts transparent triangles, whose pieces
    share a key (the tie path of the rank sort), in tiles deep enough to be split into quarters, each of which sorts
    only the part of the bin that reaches its rows.  The frame is the oracle's; whole tiles give it too."""
    for k, cam in enumerate((((6.0, 11.0, 6.02), 0.0, 1.5708), ((5.2, 12.5, -5.97), -0.2, 1.5708), ((34.0, 9.5, 6.05), 0.15, 4.7124))):
        a, b = both(T.render_sponza, hip, oracle, 960, 540, lod=1, tex_size=64, camera=cam, instrument=True)
        assert_same(a, b, f"camera {k} inside a curtain")
        c = T.render_sponza(hip, 960, 540, lod=1, tex_size=64, camera=cam, instrument=True, tuning=8)
        assert_same(a, c, f"camera {k}: quarters vs whole tiles")


def test_large_and_small_passes_alternate(hip, oracle):
    """Stage 1 of a pass of at most 4096 tiles runs on a stream of its own (highest priority, four sets deep), that of a
    larger pass on the normal one two sets deep: a 3840x2160 context that alternates full frames and row bands — unfenced,
    each band drawn over what the full frame left (colour LOAD, no clear in between) — ends with the oracle's targets."""
    def run(lib):
        r, scene, opaque, transparent = T.setup_sponza(lib, 3840, 2160, lod=8, tex_size=64)
        r.clear_color((1, 1, 1, 1))
        for k in range(6):
            if k % 2 == 0:
                r.set_scissor(0, 0, 3840, 2160)       # 8160 tiles
                r.clear_color((0.2 * k, 1.0, 1.0 - 0.1 * k, 1.0))
            else:
                r.set_scissor(0, 301 + 97 * k, 3840, 411)  # a band: ~1680 tiles, over the full frame underneath
            r.draw_geometry(scene, opaque, transparent)
        out = T._finish(r)
        r.close()
        return out
    a, b = run(hip), run(oracle)
    assert_same(a, b, "alternating pass sizes", stats=False)


def test_vertex_runs_of_every_length(hip, oracle):
    """The setup kernel reads a chunk's vertices as one run of the vertex buffer when its 192 indices name at most 128
    consecutive vertices, and gathers them per corner otherwise.  The atrium with the triangles of every surface shuffled —
    wholly (runs of thousands: the gather path) and within windows of 48 and 90 triangles (runs around the limit) — is still
    the oracle's frame (the shuffle changes submission order, for both alike)."""
    import copy
    pkg = g.load_package()
    S = pkg.scenes
    for window in (0, 48, 90):
        sc = copy.deepcopy(T.sponza_scene(8, 64))
        rng = np.random.default_rng(100 + window)
        for mesh in sc.meshes:
            idx = mesh.indices.copy()
            for sf in mesh.surfaces:
                tris = idx[sf.start_index:sf.start_index + sf.count].reshape(-1, 3)
                n = tris.shape[0]
                if window == 0:
                    tris[:] = tris[rng.permutation(n)]
                else:
                    for a in range(0, n, window):
                        b = min(n, a + window)
                        tris[a:b] = tris[a + rng.permutation(b - a)]
            mesh.indices = idx
        frames = []
        for lib in (hip, oracle):
            r = lib.create(640, 360)
            handles = sc.upload(r)
            opaque, transparent = sc.render_objects(handles)
            scene = S.scene_data_struct(*S.config3_camera(), 640, 360)
            r.set_option(1, 1)
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
            frames.append(T._finish(r))
            r.close()
        assert_same(frames[0], frames[1], f"triangles shuffled in windows of {window}")


def test_draws_that_start_anywhere(hip, oracle):
    """Wave chunks are cut along the mesh's index-group grid (64 triangles), not from a draw's first triangle: draws that
    start one triangle into a group, one short of its end, exactly on it, that are shorter than a chunk or end mid-group —
    and one whose first index is not a multiple of 3 (its triangles straddle the grid: the plain cut) — are the oracle's
    frame, on the host path and with cull / sort / draw records / chunks made on the device."""
    pkg = g.load_package()
    S, A = pkg.scenes, pkg.abi
    rng = np.random.default_rng(77)
    # one mesh: a 40 x 40 grid of quads in front of the camera, z varying, 3200 triangles, vertices near their triangles
    n = 40
    xs, ys = np.meshgrid(np.linspace(-6, 6, n + 1), np.linspace(-3.5, 3.5, n + 1))
    v = np.zeros((n + 1) * (n + 1), dtype=A.VERTEX_DTYPE)
    v["position"][:, 0], v["position"][:, 1] = xs.ravel(), ys.ravel()
    v["position"][:, 2] = -12.0 + rng.uniform(-1.5, 1.5, v.size)
    v["normal"] = (0.0, 0.6, 0.8)
    v["uv_x"], v["uv_y"] = (xs.ravel() + 6) / 3, (ys.ravel() + 3.5) / 3
    v["color"] = rng.uniform(0.4, 1.0, (v.size, 4))
    quads = np.arange(n * n)
    a = quads // n * (n + 1) + quads % n
    idx = np.stack([a, a + 1, a + n + 1, a + 1, a + n + 2, a + n + 1], axis=1).astype(np.uint32).reshape(-1)
    draws = [(0, 1), (3 * 1, 62), (3 * 63, 1), (3 * 64, 64), (3 * 128, 65), (3 * 200, 300), (3 * 511, 130), (3 * 700 + 1, 100),
             (3 * 1000, 1000), (3 * 2047, 2), (3 * 2100, 1100)]  # (first index, triangles)
    for flatten in (2, 1):
        frames = []
        for lib in (hip, oracle):
            r = lib.create(640, 360)
            mesh = r.upload_mesh(idx, v)
            img = r.create_image(S.make_texture(np.random.default_rng(5), 64, 0) if hasattr(S, "make_texture") else S.checkerboard_32(), mipmapped=True)
            smp = r.create_sampler(**S.SAMPLER_TRILINEAR)
            mats = [r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), img, smp), r.write_material(A.PASS_TRANSPARENT, (0.3, 0.3, 0.3, 1), img, smp)]
            objs = np.zeros(len(draws), dtype=A.RENDER_OBJECT_DTYPE)
            for k, (first, tris) in enumerate(draws):
                objs[k]["first_index"], objs[k]["index_count"], objs[k]["mesh"] = first, 3 * tris, mesh
                objs[k]["material"] = mats[1] if k in (3, 8) else mats[0]
                objs[k]["origin"], objs[k]["extents"], objs[k]["sphere_radius"] = (0, 0, -12), (6, 3.5, 1.5), 8.0
                objs[k]["transform"] = np.eye(4, dtype=np.float32).reshape(16)
            opaque = objs[[k for k in range(len(draws)) if k not in (3, 8)]]
            transparent = objs[[3, 8]]
            scene = S.scene_data_struct((0.0, 0.0, 0.0), 0.0, 0.0, 640, 360)
            r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
            r.set_option(A.OPT_DEVICE_FLATTEN, flatten)
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
            frames.append(T._finish(r))
            r.close()
        assert frames[1]["stats"].shaded_fragments > 20000
        assert_same(frames[0], frames[1], f"draws that start anywhere (device flatten {flatten})", stats=False)


def test_split_tiles_change_nothing(hip):
    """SVR_OPT_TUNING bit 3 keeps heavy tiles whole; the quarters of split tiles give the same frame.  (At this size
    the curtain tiles hold hundreds of transparent triangles and the pass's mean load per slot is small: they split.)"""
    a = T.render_sponza(hip, 960, 540, lod=1, tex_size=128, instrument=True)
    b = T.render_sponza(hip, 960, 540, lod=1, tex_size=128, instrument=True, tuning=8)
    assert_same(a, b, "split vs whole tiles")
    c = T.render_sponza(hip, 960, 540, lod=1, tex_size=128, scissor=(0, 100, 960, 211))
    d = T.render_sponza(hip, 960, 540, lod=1, tex_size=128, scissor=(0, 100, 960, 211), tuning=8)
    assert np.array_equal(c["color"], d["color"]) and np.array_equal(c["depth"], d["depth"])


def test_hierarchical_depth_test_changes_nothing(hip, oracle):
    """Phase A's hierarchical depth test (k_tile.hip scan_columns + the deep bins' filter) drops triangles that cannot
    win anywhere they reach.  SVR_OPT_TUNING bit 6 forces it whatever the pass's bins hold, bit 5 forbids it.
    Uninstrumented passes really drop (the frame must be the oracle's); instrumented ones walk everything — their
    fragment counts are the oracle's — and check every fragment of a triangle the test would have dropped: one that
    wins fails the pass (svr_get_stats raises)."""
    inst, cam5 = S.config5_instances(), S.config5_camera()
    cases = [dict(width=960, height=540, lod=1, tex_size=64),                                    # whole tiles and quarters, bins to ~3000
             dict(width=960, height=540, lod=1, tex_size=64, camera=((2.5, 1.0, -5.5), 0.2, 1.0)),   # inside a column: clipped pieces
             dict(width=1280, height=720, lod=2, tex_size=32, camera=cam5, instances=inst),      # 16 interpenetrating instances: deep bins, the filter's windows
             dict(width=333, height=187, lod=2, tex_size=64, scissor=(16, 8, 300, 170))]          # tiles cut by the scissor
    for kw in cases:
        ref = T.render_sponza(oracle, instrument=True, threads=16, **kw)
        for tuning in (64, 32):
            a = T.render_sponza(hip, instrument=True, tuning=tuning, **kw)   # counts + the self-check
            assert_same(a, ref, f"hierarchical depth test, tuning {tuning}, instrumented, {kw}")
            b = T.render_sponza(hip, tuning=tuning, **kw)                    # the kernel that drops
            for key in ("color", "depth", "rgba8"):
                T.assert_images_identical(b[key], ref[key], f"hierarchical depth test, tuning {tuning}, {key}, {kw}")

    # a rank of the interleaved multi-GPU form (every third tile row, svr_set_row_interleave), RGBA8 target: with and without
    frames = []
    for tuning in (64, 32):
        r, scene, opaque, transparent = T.setup_sponza(hip, 960, 540, lod=1, tex_size=64, color_format=1)
        r.set_option(4, tuning)
        r.set_row_interleave(3, 1)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        frames.append((r.read_color(), r.read_depth()))
        r.close()
    T.assert_images_identical(frames[0][0], frames[1][0], "hierarchical depth test under a row interleave, colour")
    T.assert_images_identical(frames[0][1], frames[1][1], "hierarchical depth test under a row interleave, depth")


def test_config5_instanced_reduced(hip, oracle):
    inst = S.config5_instances()
    cam = S.config5_camera()
    a = T.render_sponza(hip, 480, 270, lod=8, tex_size=32, camera=cam, instances=inst, instrument=True)
    b = T.render_sponza(oracle, 480, 270, lod=8, tex_size=32, camera=cam, instances=inst, instrument=True, threads=8)
    assert_same(a, b, "config5 reduced")
    assert a["stats"].drawcall_count > 3000


def test_camera_inside_geometry(hip, oracle):
    """Cameras that sit inside columns / look along walls: near-plane and guard-band clipping."""
    for cam in (((2.5, 1.0, -5.5), 0.2, 1.0), ((30.0, 8.0, 9.7), -0.3, 3.0), ((54.5, 15.5, 0.0), -1.2, 4.6)):
        a, b = both(T.render_sponza, hip, oracle, 256, 144, lod=4, tex_size=64, camera=cam, instrument=True)
        assert_same(a, b, f"camera {cam}")


def test_mesh_vert_stage(hip, oracle):
    sc = T.sponza_scene(8, 64)
    scene = S.scene_data_struct(*S.config3_camera(), 640, 360)
    for mesh_idx, world in sc.nodes[:12]:
        out = []
        for lib in (hip, oracle):
            r = lib.create(8, 8)
            img = r.create_image(S.white_1x1())
            smp = r.create_sampler()
            mat = r.write_material(A.PASS_MAIN_COLOR, (0.7, 0.9, 0.3, 1.0), img, smp)
            m = sc.meshes[mesh_idx]
            mh = r.upload_mesh(m.indices, m.vertices)
            out.append(r.run_mesh_vert(mh, 0, m.vertices.size, world, scene, mat))
            r.close()
        assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32)), "gl_Position bits"
        assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32)), "varyings bits"


def test_mip_generation(hip, oracle):
    rng = np.random.default_rng(17)
    for (h, w) in ((64, 64), (32, 128), (1, 1), (37, 21), (256, 256)):
        tex = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        levels = []
        for lib in (hip, oracle):
            r = lib.create(8, 8)
            img = r.create_image(tex, mipmapped=True)
            n = int(np.floor(np.log2(max(h, w)))) + 1
            levels.append([r.read_image_level(img, l) for l in range(n)])
            r.close()
        for l, (x, y) in enumerate(zip(*levels)):
            assert np.array_equal(x, y), f"{w}x{h} level {l}"


def test_deferred_shading_never_shades_more_than_forward(hip, oracle):
    a, b = both(T.render_sponza, hip, oracle, 320, 180, lod=8, tex_size=64, instrument=True)
    assert a["stats"].shaded_fragments <= b["stats"].shaded_fragments
    assert a["stats"].shaded_fragments >= int((a["depth"] > 0).sum())


def test_idempotent_and_stateless_across_frames(hip):
    """Rendering the same frame repeatedly into the same context gives the same bits (the pass
    clears its own counters / bins; depth CLEAR is part of the pass)."""
    r, scene, opaque, transparent = T.setup_sponza(hip, 640, 360, lod=4, tex_size=64)
    frames = []
    for _ in range(3):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        frames.append((r.read_color().copy(), r.read_depth().copy()))
    r.close()
    for c, d in frames[1:]:
        assert np.array_equal(c, frames[0][0]) and np.array_equal(d, frames[0][1])


def test_full_size_properties_4k(hip):
    """BASELINE configs[3] at full size: properties that need no oracle.  Band decomposition (8 bands
    of 270 rows, the 8-GPU split) reproduces the single full-frame render bit for bit; every covered
    pixel was shaded exactly once plus the transparent layers; depth lies in [0,1]."""
    W, H = 3840, 2160
    r, scene, opaque, transparent = T.setup_sponza(hip, W, H, lod=1, tex_size=256)
    r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
    r.clear_color((1, 1, 1, 1))
    st0 = r.draw_geometry(scene, opaque, transparent)
    r.sync()
    full_c, full_d = r.read_color().copy(), r.read_depth().copy()
    st = r.get_stats()
    assert st0.triangle_count > 200000
    assert 0.0 <= full_d.min() and full_d.max() <= 1.0
    assert st.shaded_fragments >= int((full_d > 0).sum())
    assert st.rasterized_fragments >= st.shaded_fragments
    tot = 0
    for k in range(8):
        r.set_scissor(0, k * 270, W, 270)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        tot += r.get_stats().rasterized_fragments
    # the last call leaves bands 0..7 all rendered by their own pass
    assert np.array_equal(r.read_color(), full_c)
    assert np.array_equal(r.read_depth(), full_d)
    assert tot == st.rasterized_fragments  # every fragment belongs to exactly one band
    r.close()


def test_full_size_properties_8k_instanced(hip):
    """BASELINE configs[4] at full size — 7680x4320, the scene x16 instances (4.2 M triangles, 5408 draws,
    device-flattened) — by properties: the frame does not depend on where cull/sort run (host vs device),
    nor on how it is cut into bands; every fragment belongs to exactly one band; an identical second frame."""
    W, H = 7680, 4320
    cam, inst = S.config5_camera(), S.config5_instances()
    r, scene, opaque, transparent = T.setup_sponza(hip, W, H, lod=1, tex_size=128, camera=cam, instances=inst)
    assert len(opaque) + len(transparent) == 5408
    r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
    frames, stats = [], []
    for mode in (1, 2, 1):  # device flatten, host path, device again
        r.set_option(A.OPT_DEVICE_FLATTEN, mode)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        st = r.get_stats()
        frames.append((r.read_color().copy(), r.read_depth().copy()))
        stats.append((st.triangle_count, st.drawcall_count, st.culled_draws, st.rasterized_fragments, st.shaded_fragments, st.binned_triangles))
    assert stats[0] == stats[1] == stats[2]
    assert stats[0][0] > 3_000_000 and stats[0][3] >= stats[0][4] > W * H // 4
    for c, d in frames[1:]:
        assert np.array_equal(c, frames[0][0]) and np.array_equal(d, frames[0][1])
    assert 0.0 <= frames[0][1].min() and frames[0][1].max() <= 1.0
    r.set_option(A.OPT_DEVICE_FLATTEN, 0)
    tot = 0
    for k in range(8):
        r.set_scissor(0, k * 540, W, 540)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        tot += r.get_stats().rasterized_fragments
    assert np.array_equal(r.read_color(), frames[0][0]) and np.array_equal(r.read_depth(), frames[0][1])
    assert tot == stats[0][3]
    r.close()


def test_row_costs_are_the_tile_cost_model_per_tile_row(hip):
    """svr_get_row_costs (what the multi-GPU form balances its row bands with): per tile row of the scissor the
    sum of 40 + opaque / 8 + 3/4 transparent over the row's tiles, posted by the pass itself."""
    W, H = 640, 360
    r, scene, opaque, transparent = T.setup_sponza(hip, W, H, lod=4, tex_size=64)
    assert r.row_costs()[0].size == 0  # nothing validated yet
    for (y0, rows) in ((0, H), (50, 201)):
        r.set_scissor(0, y0, W, rows)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
        r.sync()
        costs, cy0, crows = r.row_costs()
        op, tr = r.read_bins()
        tx, ty = (W + 31) // 32, (rows + 31) // 32
        want = (40 + (op.astype(np.int64) >> 3) + tr - (tr.astype(np.int64) >> 2)).reshape(ty, tx).sum(axis=1)
        assert (cy0, crows) == (y0, rows) and costs.size == ty
        assert np.array_equal(costs.astype(np.int64), want)
    # and the partition made from it is balanced under that model
    prof = pkg.dist.BandPlan.spread(costs, cy0, crows, H)
    b = pkg.dist.balanced_bounds(prof, 4)
    parts = [int(prof[x:y].sum()) for x, y in zip(b, b[1:])]
    assert max(parts) <= prof.sum() / 4 + prof.max() * 1
    r.close()


def test_texel_arena_grows_and_reuses(hip, oracle):
    """Every image of a context lives in one arena (texel = arena + 32-bit offset): growing it moves the texels of the
    images already there (their offsets stay), destroying an image leaves a hole the next one of its size takes, and a
    frame rendered afterwards still samples the right texels."""
    rng = np.random.default_rng(99)
    r = hip.create(64, 64)
    small = [rng.integers(0, 256, (32, 32, 4), dtype=np.uint8) for _ in range(3)]
    h_small = [r.create_image(t, mipmapped=True) for t in small]
    big = rng.integers(0, 256, (2048, 2048, 4), dtype=np.uint8)
    h_big = [r.create_image(big, mipmapped=True) for _ in range(5)]            # 5 x 22 MiB: past the first 64 MiB
    for t, h in zip(small, h_small):
        assert np.array_equal(r.read_image_level(h, 0), t)                      # survived the moves
    assert np.array_equal(r.read_image_level(h_big[0], 0), big) and np.array_equal(r.read_image_level(h_big[4], 0), big)
    lvl3 = r.read_image_level(h_big[2], 3)
    r.destroy_image(h_big[1])
    again = r.create_image(big[::-1].copy(), mipmapped=True)                    # takes the hole
    assert np.array_equal(r.read_image_level(again, 0), big[::-1]) and np.array_equal(r.read_image_level(h_big[2], 3), lvl3)
    assert np.array_equal(r.read_image_level(h_big[0], 0), big)
    r.close()
    # and a frame over textures created around a growth step is the oracle's
    def frame(lib):
        rr = lib.create(96, 64)
        pad = [rr.create_image(big, mipmapped=True) for _ in range(2)]
        tex = rr.create_image(small[0], mipmapped=True)
        pad.append(rr.create_image(big, mipmapped=True))                       # grows the HIP arena after `tex` was placed
        mesh = rr.upload_mesh(SC.QUAD_IDX, SC.clip_quad(-1, -1, 1, 1, 0.5))
        smp = rr.create_sampler(**S.SAMPLER_TRILINEAR)
        mat = rr.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), tex, smp)
        rr.clear_color((0, 0, 0, 1))
        rr.draw_geometry(SC.identity_scene(), SC.objs([SC.render_object(mesh, mat, 0, 6)]))
        out = rr.read_color().copy()
        rr.close()
        return out
    assert np.array_equal(frame(hip), frame(oracle))


def test_reciprocal_all_inputs(hip):
    """The contract's "IEEE 1/x" (perspective divide, 1/area, 1/q per fragment) is computed without the
    compiler's division expansion: v_rcp_f32 + one Newton step inside an exponent window, the division
    proper outside it.  All 2^32 bit patterns against the compiler's correctly rounded 1.0f / x."""
    r = hip.create(16, 16)
    bad, refined, pats = r.rcp_sweep(0)
    assert bad == 0, [hex(int(p)) for p in pats]
    assert refined == 2 * 191 * (1 << 23)   # both signs of the 191 binades of the window took the division-free path
    bad1, _, pats1 = r.rcp_sweep(1)          # the refinement alone, never the fallback, inside the window
    assert bad1 == 0, [hex(int(p)) for p in pats1]
    r.close()


def test_errors_on_the_hip_library(hip):
    r = hip.create(16, 16)
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
        r.set_scissor(8, 8, 16, 4)
    with pytest.raises(pkg.SvrError):
        hip.create(0, 16)
    r.close()


def test_bound_targets_and_stream(hip, oracle):
    """Render into caller-owned device memory (torch tensors) on a caller-owned stream."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    W, H = 320, 180
    r, scene, opaque, transparent = T.setup_sponza(hip, W, H, lod=8, tex_size=64)
    color = torch.zeros((H, W, 4), dtype=torch.float16, device=dev)
    depth = torch.full((H, W), 7.0, dtype=torch.float32, device=dev)
    s = torch.cuda.Stream(device=dev)
    r.set_stream(s.cuda_stream)
    r.bind_targets(color.data_ptr(), depth.data_ptr())
    r.clear_color((1, 1, 1, 1))
    r.draw_geometry(scene, opaque, transparent)
    r.sync()
    ref = T.render_sponza(oracle, W, H, lod=8, tex_size=64)
    assert np.array_equal(color.cpu().numpy().view(np.uint16), ref["color"])
    assert np.array_equal(depth.cpu().numpy(), ref["depth"])
    r.close()
