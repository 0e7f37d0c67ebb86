"""The BASELINE workloads at their own sizes, with the textures bench.py uses (25 x 1024^2, mip-mapped):
tests/golden/full_frames.json holds SHA-256 digests of the oracle's frames (made in the container by
tests/golden/make_full_frames.py); the HIP frames must hash to the same values.  The oracle itself is
run on the GPU box only over three 32-row scissor bands of the 4K frame (its cost is proportional to rows).
"""
import importlib.util
import json
import os

import numpy as np
import pytest

import __graft_entry__ as g
import svr_testlib as T

_spec = importlib.util.spec_from_file_location("make_full_frames", os.path.join(T.GOLDEN_DIR, "make_full_frames.py"))
MF = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(MF)

pkg = g.load_package()
A, S = pkg.abi, pkg.scenes


def golden():
    with open(MF.OUT) as f:
        return json.load(f)


def test_goldens_cover_every_full_size_config():
    doc = golden()
    for name, (w, h, _) in MF.FRAMES.items():
        d = doc[name]
        for key in ("color", "depth", "rgba8"):
            assert len(d[key]) == 64 and len(d[key + "_strips"]) == doc["strips"] == MF.STRIPS
        c = d["counters"]
        assert c["triangle_count"] > 200000 and c["rasterized_fragments"] >= c["covered_pixels"] > w * h // 3
    # configs[4] as SURVEY §8d specifies it: the whole frame covered, depth complexity >= 8
    c = doc["config4_x16_7680x4320"]["counters"]
    assert c["covered_pixels"] == 7680 * 4320
    assert c["rasterized_fragments"] >= 8 * c["covered_pixels"]
    assert c["drawcall_count"] > 5000


def test_oracle_still_reproduces_a_strip_of_the_4k_golden(oracle):
    """One of the sixteen strips of configs[3], re-rendered by the oracle under a scissor (a band of a
    frame is that band of the full frame): freezes oracle, scene generator and texture path together."""
    doc = golden()
    assert doc["scene_sha256"] == MF.scene_fingerprint(), "the seeded scene generator no longer produces the bytes the goldens were made from"
    w, h, _ = MF.FRAMES["config3_3840x2160"]
    k = 9
    y0, y1 = MF.strip_rows(h, k)
    out = T.render_sponza(oracle, w, h, lod=1, tex_size=MF.TEX, scissor=(0, y0, w, y1 - y0), threads=len(os.sched_getaffinity(0)))
    d = doc["config3_3840x2160"]
    for key in ("color", "depth", "rgba8"):
        assert MF.sha(out[key][y0:y1]) == d[key + "_strips"][k], f"{key} rows {y0}..{y1}"


def _compare(out, d, name, height):
    got = MF.digest(out, height)
    for key in ("color", "depth", "rgba8"):
        bad = [k for k in range(MF.STRIPS) if got[key + "_strips"][k] != d[key + "_strips"][k]]
        assert not bad, f"{name}: {key} differs from the oracle's frame in strips {bad} (rows {[MF.strip_rows(height, k) for k in bad]})"
        assert got[key] == d[key], f"{name}: {key} digest"
    assert got["counters"] == d["counters"], f"{name}: counters"


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(MF.FRAMES))
def test_hip_frame_hashes_to_the_oracle_frame(hip, name):
    """configs[2], configs[3] (the frame the headline number is quoted on) and configs[4], every pixel."""
    doc = golden()
    out = MF.render(hip, name)
    _compare(out, doc[name], name, MF.FRAMES[name][1])


@pytest.mark.gpu
def test_device_flatten_and_bands_hash_to_the_same_8k_frame(hip):
    """configs[4] again the way eight ranks render it: host-side cull/sort instead of the device pass, and
    eight row bands under their own scissor into one target."""
    doc = golden()
    name = "config4_x16_7680x4320"
    w, h, _ = MF.FRAMES[name]
    r, scene, opaque, transparent = T.setup_sponza(hip, w, h, lod=1, tex_size=MF.TEX, camera=S.config5_camera(),
                                                   instances=S.config5_instances())
    r.set_option(A.OPT_DEVICE_FLATTEN, 2)
    for k in range(8):
        r.set_scissor(0, k * (h // 8), w, h // 8)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
    out = T._finish(r)
    r.close()
    d = doc[name]
    for key in ("color", "depth", "rgba8"):
        assert MF.sha(out[key]) == d[key], f"banded {key}"


@pytest.mark.gpu
def test_three_bands_of_the_4k_frame_against_the_oracle(hip, oracle):
    """HIP == oracle, bit for bit, on 32-row bands of the exact bench frame (3840x2160, 1024^2 textures):
    the busiest rows (curtains, columns), the floor and the ceiling."""
    w, h = 3840, 2160
    bands = (1040, 1700, 300)
    res = {}
    for lib in (hip, oracle):
        r, scene, opaque, transparent = T.setup_sponza(lib, w, h, lod=1, tex_size=MF.TEX)
        if lib.backend == "cpu-oracle":
            lib.lib.svr_oracle_set_threads(r.h, min(16, len(os.sched_getaffinity(0))))
        r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
        rows = []
        for y0 in bands:
            r.set_scissor(0, y0, w, 32)
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
            r.sync()
            st = r.get_stats()
            rows.append((r.read_color()[y0:y0 + 32].copy(), r.read_depth()[y0:y0 + 32].copy(),
                         r.read_color(as_rgba8=True)[y0:y0 + 32].copy(), int(st.rasterized_fragments), int(st.binned_triangles)))
        res[lib.backend] = rows
        r.close()
    for y0, a, b in zip(bands, res["hip-gfx950"], res["cpu-oracle"]):
        T.assert_images_identical(a[0], b[0], f"rows {y0}.. colour")
        T.assert_images_identical(a[1], b[1], f"rows {y0}.. depth")
        T.assert_images_identical(a[2], b[2], f"rows {y0}.. rgba8")
        assert a[3:] == b[3:], f"rows {y0}.. counters"
        assert (a[1] > 0).mean() > 0.9
