"""Test-side helpers: loads the CPU oracle (the checker) and renders the BASELINE configs through
any library that exports the svr.h ABI.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product package never does."""
import ctypes as C
import functools
import os
import subprocess

import numpy as np

import __graft_entry__ as g

ROOT = g.ROOT
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libsvr_oracle.so")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


@functools.lru_cache(maxsize=None)
def load_oracle():
    # (re)build on this machine: the Makefile keys the build on the host's CPU flags
    subprocess.run(["make", "-s"], cwd=ORACLE_DIR, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    pkg = g.load_package()
    lib = pkg.SvrLib(ORACLE_LIB)
    assert lib.backend == "cpu-oracle"
    lib.lib.svr_oracle_set_threads.argtypes = [C.c_void_p, C.c_int]
    lib.lib.svr_oracle_is_visible.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.lib.svr_oracle_f32_to_f16.argtypes = [C.c_float]
    lib.lib.svr_oracle_f32_to_f16.restype = C.c_uint16
    lib.lib.svr_oracle_f16_to_f32.argtypes = [C.c_uint16]
    lib.lib.svr_oracle_f16_to_f32.restype = C.c_float
    lib.lib.svr_oracle_lod.argtypes = [C.c_float]
    lib.lib.svr_oracle_lod.restype = C.c_float
    return lib


def _finish(r, stats=None):
    r.sync()
    out = {"color": r.read_color(), "depth": r.read_depth(), "rgba8": r.read_color(as_rgba8=True),
           "stats": r.get_stats()}
    return out


def render_config1(lib, size=256, color_format=0, instrument=False):
    """colored_triangle.vert/.frag over a white background (BASELINE config 1)."""
    r = lib.create(size, size, color_format)
    r.set_option(1, 1 if instrument else 0)
    r.clear_color((1, 1, 1, 1))
    r.draw_colored_triangle()
    out = _finish(r)
    r.close()
    return out


def render_config2(lib, width=1920, height=1080, color_format=0, instrument=False):
    """textured cube through colored_triangle_mesh.vert + tex_image.frag (BASELINE config 2)."""
    pkg = g.load_package()
    S = pkg.scenes
    r = lib.create(width, height, color_format)
    r.set_option(1, 1 if instrument else 0)
    cube = S.cube_mesh()
    mesh = r.upload_mesh(cube.indices, cube.vertices)
    img = r.create_image(S.checkerboard_32(), mipmapped=False)
    smp = r.create_sampler(**S.SAMPLER_NEAREST)
    r.clear_color((1, 1, 1, 1))
    r.draw_tex_image(mesh, 0, cube.indices.size, S.config2_render_matrix(width, height), img, smp)
    out = _finish(r)
    r.close()
    return out


@functools.lru_cache(maxsize=4)
def sponza_scene(lod, tex_size):
    return g.load_package().scenes.sponza_like(lod=lod, tex_size=tex_size)


def setup_sponza(lib, width, height, lod=8, tex_size=64, color_format=0, window=None, camera=None,
                 instances=None):
    pkg = g.load_package()
    S = pkg.scenes
    sc = sponza_scene(lod, tex_size)
    r = lib.create(width, height, color_format)
    handles = sc.upload(r)
    opaque, transparent = sc.render_objects(handles, instance_transforms=instances)
    pos, pitch, yaw = camera if camera is not None else S.config3_camera()
    ww, wh = window if window is not None else (width, height)
    scene = S.scene_data_struct(pos, pitch, yaw, ww, wh)
    return r, scene, opaque, transparent


def render_sponza(lib, width, height, lod=8, tex_size=64, color_format=0, scissor=None, camera=None,
                  instances=None, threads=None, instrument=False, trace=None, queue_caps=None, device_flatten=None,
                  tuning=None):
    """mesh.vert/mesh.frag over the synthetic atrium (BASELINE configs 3-5 at reduced size)."""
    r, scene, opaque, transparent = setup_sponza(lib, width, height, lod, tex_size, color_format,
                                                 camera=camera, instances=instances)
    if threads and lib.backend == "cpu-oracle":
        lib.lib.svr_oracle_set_threads(r.h, threads)
    r.set_option(1, 1 if (instrument or trace) else 0)
    if queue_caps is not None:
        r.set_option(5, queue_caps)  # SVR_OPT_QUEUE_CAPS
    if device_flatten is not None:
        r.set_option(6, device_flatten)  # SVR_OPT_DEVICE_FLATTEN
    if tuning is not None:
        r.set_option(4, tuning)  # SVR_OPT_TUNING
    if trace:
        r.trace_pixel(*trace)
    r.clear_color((1, 1, 1, 1))
    if scissor is not None:
        r.set_scissor(*scissor)
    r.draw_geometry(scene, opaque, transparent)
    out = _finish(r)
    out["n_opaque"], out["n_transparent"] = len(opaque), len(transparent)
    if trace:
        out["trace"] = r.read_trace()
    r.close()
    return out


def f16_bits_to_f32(a):
    return a.view(np.float16).astype(np.float32)


def assert_images_identical(a, b, what):
    if a.shape != b.shape or not np.array_equal(a.view(np.uint8), b.view(np.uint8)):
        diff = (a != b)
        n = int(diff.sum())
        idx = np.argwhere(diff)[:5]
        raise AssertionError(f"{what}: {n} of {a.size} elements differ, first at {idx.tolist()}: "
                             f"{[a[tuple(i)] for i in idx]} vs {[b[tuple(i)] for i in idx]}")
