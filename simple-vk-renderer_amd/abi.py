"""ctypes mirror of include/svr.h and a thin object wrapper over one loaded library.

The same binding drives the product (libsvr_hip.so) and, in tests only, the CPU oracle's library:
both export the identical C ABI.  Nothing here picks a library by itself; the product is loaded by
__init__.load_product_library(), the oracle only by the helpers under tests/.
"""
import ctypes as C
import os

import numpy as np


class SvrVertex(C.Structure):  # src/vk_types.h:97-103
    _fields_ = [("position", C.c_float * 3), ("uv_x", C.c_float), ("normal", C.c_float * 3),
                ("uv_y", C.c_float), ("color", C.c_float * 4)]


class SvrSceneData(C.Structure):  # src/vk_types.h:118-125
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("viewproj", C.c_float * 16),
                ("ambient_color", C.c_float * 4), ("sunlight_direction", C.c_float * 4),
                ("sunlight_color", C.c_float * 4)]


class SvrBounds(C.Structure):  # src/vk_loader.h:11-15
    _fields_ = [("origin", C.c_float * 3), ("sphere_radius", C.c_float), ("extents", C.c_float * 3)]


class SvrRenderObject(C.Structure):  # src/vk_engine.h:29-38
    _fields_ = [("index_count", C.c_uint32), ("first_index", C.c_uint32), ("mesh", C.c_uint32),
                ("material", C.c_uint32), ("bounds", SvrBounds), ("transform", C.c_float * 16)]


class SvrSamplerDesc(C.Structure):
    _fields_ = [("mag_filter", C.c_int32), ("min_filter", C.c_int32), ("mipmap_mode", C.c_int32),
                ("min_lod", C.c_float), ("max_lod", C.c_float)]


class SvrStats(C.Structure):  # src/vk_engine.h:16-22 + extensions
    _fields_ = [("frame_time", C.c_float), ("triangle_count", C.c_int32), ("drawcall_count", C.c_int32),
                ("scene_update_time", C.c_float), ("mesh_draw_time", C.c_float),
                ("gpu_time_ms", C.c_float), ("culled_draws", C.c_uint32), ("timed_passes", C.c_uint32),
                ("shaded_fragments", C.c_uint64), ("rasterized_fragments", C.c_uint64),
                ("binned_triangles", C.c_uint64), ("bin_entries", C.c_uint64),
                ("geometry_ms", C.c_float), ("binning_ms", C.c_float), ("tile_ms", C.c_float),
                ("replayed_passes", C.c_uint32)]


class SvrConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("color_format", C.c_int32), ("reserved", C.c_uint32 * 4)]


VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("uv_x", "<f4"), ("normal", "<f4", 3),
                         ("uv_y", "<f4"), ("color", "<f4", 4)])
RENDER_OBJECT_DTYPE = np.dtype([("index_count", "<u4"), ("first_index", "<u4"), ("mesh", "<u4"),
                                ("material", "<u4"), ("origin", "<f4", 3), ("sphere_radius", "<f4"),
                                ("extents", "<f4", 3), ("transform", "<f4", 16)])
assert VERTEX_DTYPE.itemsize == C.sizeof(SvrVertex) == 48
assert RENDER_OBJECT_DTYPE.itemsize == C.sizeof(SvrRenderObject) == 108
assert C.sizeof(SvrSceneData) == 240
assert C.sizeof(SvrBounds) == 28

COLOR_RGBA16F, COLOR_RGBA8 = 0, 1
PASS_MAIN_COLOR, PASS_TRANSPARENT, PASS_OTHER = 0, 1, 2
FILTER_NEAREST, FILTER_LINEAR = 0, 1
MIPMAP_NEAREST, MIPMAP_LINEAR = 0, 1
LOD_CLAMP_NONE = 1000.0
OPT_COUNT_FRAGMENTS = 1
OPT_KERNEL_TIMING = 2
OPT_TILE_CYCLES = 3
OPT_TUNING = 4
OPT_QUEUE_CAPS = 5
OPT_DEVICE_FLATTEN = 6
BACKGROUND_GRADIENT, BACKGROUND_SKY = 0, 1
VS_COLORED_TRIANGLE, VS_COLORED_TRIANGLE_MESH = 1, 2
SWAPCHAIN_B8G8R8A8, SWAPCHAIN_R8G8B8A8 = 0, 1
GRADIENT_DEFAULT = (1.0, 1.0, 1.0, 1.0) * 2 + (0.0,) * 8       # src/vk_engine.cpp:981-982
SKY_DEFAULT = (0.1, 0.2, 0.4, 0.97) + (0.0,) * 12               # src/vk_engine.cpp:988

# every symbol include/svr.h declares
SYMBOLS = ["svr_create", "svr_destroy", "svr_set_stream", "svr_bind_targets", "svr_get_targets",
           "svr_upload_mesh", "svr_destroy_mesh", "svr_create_image", "svr_destroy_image",
           "svr_read_image_level", "svr_create_sampler", "svr_write_material", "svr_clear_color",
           "svr_draw_background", "svr_copy_to_swapchain", "svr_read_swapchain",
           "svr_set_scissor", "svr_set_row_interleave", "svr_set_present_status", "svr_draw_geometry", "svr_draw_colored_triangle", "svr_draw_tex_image",
           "svr_run_mesh_vert", "svr_run_vertex_shader", "svr_set_option", "svr_debug_trace_pixel", "svr_debug_read_trace", "svr_debug_read_bins", "svr_debug_read_tile_cycles", "svr_debug_rcp_sweep", "svr_get_row_costs", "svr_sync", "svr_read_color", "svr_read_depth", "svr_get_stats",
           "svr_last_error", "svr_backend_name"]


class SvrError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"svr error {code}: {text}")
        self.code = code


class SvrLib:
    """One loaded shared library exporting the svr.h ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        P = C.c_void_p
        L.svr_last_error.restype = C.c_char_p
        L.svr_backend_name.restype = C.c_char_p
        L.svr_destroy.restype = None
        L.svr_create.argtypes = [C.POINTER(SvrConfig), C.POINTER(P)]
        L.svr_destroy.argtypes = [P]
        L.svr_set_stream.argtypes = [P, P]
        L.svr_bind_targets.argtypes = [P, P, P]
        L.svr_get_targets.argtypes = [P, C.POINTER(P), C.POINTER(P)]
        L.svr_upload_mesh.argtypes = [P, P, C.c_size_t, P, C.c_size_t, C.POINTER(C.c_uint32)]
        L.svr_destroy_mesh.argtypes = [P, C.c_uint32]
        L.svr_create_image.argtypes = [P, P, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint32)]
        L.svr_destroy_image.argtypes = [P, C.c_uint32]
        L.svr_read_image_level.argtypes = [P, C.c_uint32, C.c_uint32, P, C.c_size_t,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.svr_create_sampler.argtypes = [P, C.POINTER(SvrSamplerDesc), C.POINTER(C.c_uint32)]
        L.svr_write_material.argtypes = [P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                         C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.svr_clear_color.argtypes = [P, C.POINTER(C.c_float)]
        L.svr_draw_background.argtypes = [P, C.c_int, C.POINTER(C.c_float)]
        L.svr_copy_to_swapchain.argtypes = [P, P, C.c_uint32, C.c_uint32, C.c_int]
        L.svr_read_swapchain.argtypes = [P, C.c_uint32, C.c_uint32, C.c_int, P, C.c_size_t]
        L.svr_set_scissor.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        if hasattr(L, "svr_set_row_interleave"):  # tools/ab_libs.py also loads builds that predate these
            L.svr_set_row_interleave.argtypes = [P, C.c_uint32, C.c_uint32]
            L.svr_set_present_status.argtypes = [P, P]
        L.svr_draw_geometry.argtypes = [P, C.POINTER(SvrSceneData), P, C.c_size_t, P, C.c_size_t,
                                        C.POINTER(SvrStats)]
        L.svr_draw_colored_triangle.argtypes = [P, C.POINTER(SvrStats)]
        L.svr_draw_tex_image.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                         C.c_uint32, C.c_uint32, C.POINTER(SvrStats)]
        L.svr_run_mesh_vert.argtypes = [P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float),
                                        C.POINTER(SvrSceneData), C.c_uint32, P, P]
        if hasattr(L, "svr_run_vertex_shader"):  # tools/ab_libs.py also loads builds that predate it
            L.svr_run_vertex_shader.argtypes = [P, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), P, P]
        L.svr_set_option.argtypes = [P, C.c_int, C.c_int64]
        L.svr_debug_trace_pixel.argtypes = [P, C.c_int, C.c_int]
        L.svr_debug_read_trace.argtypes = [P, P]
        L.svr_debug_read_bins.argtypes = [P, P, C.c_size_t, C.POINTER(C.c_uint32)]
        L.svr_debug_read_tile_cycles.argtypes = [P, P, C.c_size_t]
        if hasattr(L, "svr_get_row_costs"):
            L.svr_get_row_costs.argtypes = [P, P, C.c_size_t, P, P, P]
        if hasattr(L, "svr_debug_rcp_sweep"):  # tools/ab_libs.py also loads builds that predate it
            L.svr_debug_rcp_sweep.argtypes = [P, C.c_int, C.c_uint64, C.c_uint64, P, P, P]
        L.svr_sync.argtypes = [P]
        L.svr_read_color.argtypes = [P, P, C.c_size_t, C.c_int]
        L.svr_read_depth.argtypes = [P, P, C.c_size_t]
        L.svr_get_stats.argtypes = [P, C.POINTER(SvrStats)]

    @property
    def backend(self):
        return self.lib.svr_backend_name().decode()

    def check(self, rc):
        if rc != 0:
            raise SvrError(rc, self.lib.svr_last_error().decode(errors="replace"))

    def create(self, width, height, color_format=COLOR_RGBA16F, device=0):
        return Renderer(self, width, height, color_format, device)


def _f4(a):
    return (C.c_float * 4)(*[float(x) for x in a])


def _f16m(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(16))
    return (C.c_float * 16)(*a.tolist())


def scene_struct(view, proj, viewproj, ambient, sun_dir, sun_color):
    s = SvrSceneData()
    s.view = _f16m(view)
    s.proj = _f16m(proj)
    s.viewproj = _f16m(viewproj)
    s.ambient_color = _f4(ambient)
    s.sunlight_direction = _f4(sun_dir)
    s.sunlight_color = _f4(sun_color)
    return s


class Renderer:
    """A context of one SvrLib: the VulkanEngine-shaped surface of the draw path."""

    def __init__(self, lib, width, height, color_format=COLOR_RGBA16F, device=0):
        self.lib = lib
        self.width, self.height, self.color_format = int(width), int(height), int(color_format)
        cfg = SvrConfig(device=device, width=width, height=height, color_format=color_format)
        h = C.c_void_p()
        lib.check(lib.lib.svr_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._clear_args, self._object_args = {}, {}

    def close(self):
        if self.h:
            self.lib.lib.svr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- resources
    def upload_mesh(self, indices, vertices):
        idx = np.ascontiguousarray(indices, dtype=np.uint32)
        vtx = np.ascontiguousarray(vertices, dtype=VERTEX_DTYPE)
        out = C.c_uint32()
        self.lib.check(self.lib.lib.svr_upload_mesh(self.h, idx.ctypes.data, idx.size, vtx.ctypes.data,
                                                    vtx.size, C.byref(out)))
        return out.value

    def destroy_mesh(self, mesh):
        self.lib.check(self.lib.lib.svr_destroy_mesh(self.h, mesh))

    def create_image(self, rgba8, mipmapped=False):
        a = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert a.ndim == 3 and a.shape[2] == 4
        out = C.c_uint32()
        self.lib.check(self.lib.lib.svr_create_image(self.h, a.ctypes.data, a.shape[1], a.shape[0],
                                                     1 if mipmapped else 0, C.byref(out)))
        return out.value

    def destroy_image(self, image):
        self.lib.check(self.lib.lib.svr_destroy_image(self.h, image))

    def read_image_level(self, image, level):
        w, h = C.c_uint32(), C.c_uint32()
        self.lib.check(self.lib.lib.svr_read_image_level(self.h, image, level, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 4), dtype=np.uint8)
        self.lib.check(self.lib.lib.svr_read_image_level(self.h, image, level, out.ctypes.data, out.nbytes,
                                                         C.byref(w), C.byref(h)))
        return out

    def create_sampler(self, mag=FILTER_NEAREST, minf=FILTER_NEAREST, mip=MIPMAP_NEAREST, min_lod=0.0,
                       max_lod=0.0):
        d = SvrSamplerDesc(mag, minf, mip, min_lod, max_lod)
        out = C.c_uint32()
        self.lib.check(self.lib.lib.svr_create_sampler(self.h, C.byref(d), C.byref(out)))
        return out.value

    def write_material(self, pass_type, color_factors, image, sampler, metal_rough=(1.0, 0.5, 0.0, 0.0)):
        out = C.c_uint32()
        self.lib.check(self.lib.lib.svr_write_material(self.h, pass_type, _f4(color_factors), _f4(metal_rough),
                                                       image, sampler, C.byref(out)))
        return out.value

    # -- per frame
    def set_stream(self, stream_handle):
        self.lib.check(self.lib.lib.svr_set_stream(self.h, C.c_void_p(stream_handle)))

    def bind_targets(self, color_ptr, depth_ptr):
        self.lib.check(self.lib.lib.svr_bind_targets(self.h, C.c_void_p(color_ptr), C.c_void_p(depth_ptr)))

    def get_targets(self):
        c, d = C.c_void_p(), C.c_void_p()
        self.lib.check(self.lib.lib.svr_get_targets(self.h, C.byref(c), C.byref(d)))
        return c.value, d.value

    def clear_color(self, rgba=(1.0, 1.0, 1.0, 1.0)):
        key = tuple(rgba)
        arr = self._clear_args.get(key)  # the same few colours frame after frame: keep their ctypes arrays
        if arr is None:
            if len(self._clear_args) > 16:
                self._clear_args.clear()
            arr = self._clear_args[key] = _f4(rgba)
        self.lib.check(self.lib.lib.svr_clear_color(self.h, arr))

    def draw_background(self, effect, data):
        """effect: BACKGROUND_GRADIENT / BACKGROUND_SKY; data: the 16 floats of ComputePushConstants."""
        arr = (C.c_float * 16)(*[float(v) for v in data])
        self.lib.check(self.lib.lib.svr_draw_background(self.h, int(effect), arr))

    def copy_to_swapchain(self, dst_ptr, width, height, fmt=0):
        self.lib.check(self.lib.lib.svr_copy_to_swapchain(self.h, C.c_void_p(dst_ptr), width, height, fmt))

    def read_swapchain(self, width, height, fmt=0):
        out = np.empty((height, width, 4), dtype=np.uint8)
        self.lib.check(self.lib.lib.svr_read_swapchain(self.h, width, height, fmt, out.ctypes.data, out.nbytes))
        return out

    def set_scissor(self, x, y, w, h):
        self.lib.check(self.lib.lib.svr_set_scissor(self.h, x, y, w, h))

    def set_row_interleave(self, stride, offset):
        """of the scissor's 32-row tile rows render those with index % stride == offset (1, 0: all)"""
        self.lib.check(self.lib.lib.svr_set_row_interleave(self.h, stride, offset))

    def set_present_status(self, status_ptr):
        """device word (oracle: host word) every copy_to_swapchain reports to: 1 = void, awaiting the replay"""
        self.lib.check(self.lib.lib.svr_set_present_status(self.h, C.c_void_p(status_ptr)))

    def _objects(self, a):
        """(address, count) of a RenderObject list; arrays of the right kind are passed as they are, and their address is
        remembered (ndarray.ctypes builds an object per access: microseconds that count against a 50-us band)."""
        if a is None:
            return 0, 0
        if not (isinstance(a, np.ndarray) and a.dtype == RENDER_OBJECT_DTYPE and a.flags.c_contiguous):
            a = np.ascontiguousarray(a, dtype=RENDER_OBJECT_DTYPE)
            self._keep = (getattr(self, "_keep", ()) + (a,))[-2:]  # alive for the duration of the call
            return a.ctypes.data, a.size
        hit = self._object_args.get(id(a))
        if hit is None or hit[0] is not a:
            if len(self._object_args) > 8:
                self._object_args.clear()
            hit = self._object_args[id(a)] = (a, a.ctypes.data, a.size)
        return hit[1], hit[2]

    def draw_geometry(self, scene, opaque, transparent=None):
        op, n_op = self._objects(opaque)
        tr, n_tr = self._objects(transparent)
        st = SvrStats()
        self.lib.check(self.lib.lib.svr_draw_geometry(self.h, C.byref(scene), op, n_op, tr, n_tr, C.byref(st)))
        return st

    def draw_colored_triangle(self):
        st = SvrStats()
        self.lib.check(self.lib.lib.svr_draw_colored_triangle(self.h, C.byref(st)))
        return st

    def draw_tex_image(self, mesh, first_index, index_count, render_matrix, image, sampler):
        st = SvrStats()
        self.lib.check(self.lib.lib.svr_draw_tex_image(self.h, mesh, first_index, index_count,
                                                       _f16m(render_matrix), image, sampler, C.byref(st)))
        return st

    def run_mesh_vert(self, mesh, first_vertex, n_vertices, world, scene, material):
        clip = np.empty((n_vertices, 4), dtype=np.float32)
        var = np.empty((n_vertices, 8), dtype=np.float32)
        self.lib.check(self.lib.lib.svr_run_mesh_vert(self.h, mesh, first_vertex, n_vertices, _f16m(world),
                                                      C.byref(scene), material, clip.ctypes.data,
                                                      var.ctypes.data))
        return clip, var

    def run_vertex_shader(self, shader, mesh=0, first_vertex=0, n_vertices=3, render_matrix=None):
        """shader: VS_COLORED_TRIANGLE / VS_COLORED_TRIANGLE_MESH -> (clip [n,4], varyings [n,8])"""
        clip = np.empty((n_vertices, 4), dtype=np.float32)
        var = np.empty((n_vertices, 8), dtype=np.float32)
        mat = _f16m(render_matrix) if render_matrix is not None else None
        self.lib.check(self.lib.lib.svr_run_vertex_shader(self.h, shader, mesh, first_vertex, n_vertices, mat,
                                                          clip.ctypes.data, var.ctypes.data))
        return clip, var

    def set_option(self, option, value):
        self.lib.check(self.lib.lib.svr_set_option(self.h, option, value))

    def trace_pixel(self, x, y):
        self.lib.check(self.lib.lib.svr_debug_trace_pixel(self.h, int(x), int(y)))

    def read_bins(self):
        """(opaque counts, transparent counts) per 32x32 tile of the last pass, row-major."""
        n = C.c_uint32()
        self.lib.check(self.lib.lib.svr_debug_read_bins(self.h, None, 0, C.byref(n)))
        out = np.zeros(2 * n.value, dtype=np.uint32)
        self.lib.check(self.lib.lib.svr_debug_read_bins(self.h, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value], out[n.value:]

    def read_tile_cycles(self):
        n = C.c_uint32()
        self.lib.check(self.lib.lib.svr_debug_read_bins(self.h, None, 0, C.byref(n)))
        out = np.zeros((n.value, 4), dtype=np.uint32)
        self.lib.check(self.lib.lib.svr_debug_read_tile_cycles(self.h, out.ctypes.data, out.size))
        return out

    def row_costs(self):
        """(costs per tile row [uint32], first scissor row, scissor rows) of the last validated pass; empty before any."""
        n, y0, rows = C.c_uint32(), C.c_uint32(), C.c_uint32()
        out = np.zeros(512, dtype=np.uint32)
        self.lib.check(self.lib.lib.svr_get_row_costs(self.h, out.ctypes.data, out.size, C.byref(n), C.byref(y0), C.byref(rows)))
        return out[:n.value].copy(), y0.value, rows.value

    def rcp_sweep(self, variant=0, first=0, count=1 << 32):
        """(mismatches, inputs on the refined path, first mismatching bit patterns) of svr_debug_rcp_sweep."""
        bad, fast = C.c_uint64(), C.c_uint64()
        pats = np.zeros(16, dtype=np.uint32)
        self.lib.check(self.lib.lib.svr_debug_rcp_sweep(self.h, variant, first, count, C.byref(bad), C.byref(fast), pats.ctypes.data))
        return bad.value, fast.value, pats[:min(bad.value, 16)]

    def read_trace(self):
        out = np.zeros(64, dtype=np.float32)
        self.lib.check(self.lib.lib.svr_debug_read_trace(self.h, out.ctypes.data))
        return out

    def sync(self):
        self.lib.check(self.lib.lib.svr_sync(self.h))

    def read_color(self, as_rgba8=False):
        if as_rgba8 or self.color_format == COLOR_RGBA8:
            out = np.empty((self.height, self.width, 4), dtype=np.uint8)
        else:
            out = np.empty((self.height, self.width, 4), dtype=np.uint16)  # fp16 bit patterns
        self.lib.check(self.lib.lib.svr_read_color(self.h, out.ctypes.data, out.nbytes, 1 if as_rgba8 else 0))
        return out

    def read_depth(self):
        out = np.empty((self.height, self.width), dtype=np.float32)
        self.lib.check(self.lib.lib.svr_read_depth(self.h, out.ctypes.data, out.nbytes))
        return out

    def get_stats(self):
        st = SvrStats()
        self.lib.check(self.lib.lib.svr_get_stats(self.h, C.byref(st)))
        return st
