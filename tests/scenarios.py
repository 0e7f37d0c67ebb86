"""Small hand-built scenes rendered through the svr.h ABI.  Each scenario takes a library (oracle or
HIP) and returns {"color": fp16 bits [H,W,4], "depth": f32 [H,W], "rgba8", "stats"}.  The oracle KATs
assert analytic properties of these; the GPU parity tests assert HIP == oracle bit for bit."""
import numpy as np

import __graft_entry__ as g
import svr_testlib as T

pkg = g.load_package()
A, S, GL = pkg.abi, pkg.scenes, pkg.glmath
f32 = np.float32
IDENT = GL.identity()


def identity_scene(ambient=0.1, sun=(0, 1, 0.5, 1), sun_color=(1, 1, 1, 1)):
    """GPUSceneData with viewproj = I: vertex positions are clip coordinates."""
    return A.scene_struct(IDENT, IDENT, IDENT, [ambient] * 4, sun, sun_color)


def make_vertices(positions, normals=None, uvs=None, colors=None):
    n = len(positions)
    v = np.zeros(n, dtype=A.VERTEX_DTYPE)
    v["position"] = np.asarray(positions, dtype=f32)
    v["normal"] = np.asarray(normals, dtype=f32) if normals is not None else np.array([0, 1, 0], dtype=f32)
    v["color"] = np.asarray(colors, dtype=f32) if colors is not None else f32(1)
    if uvs is not None:
        uvs = np.asarray(uvs, dtype=f32)
        v["uv_x"], v["uv_y"] = uvs[:, 0], uvs[:, 1]
    return v


def render_object(mesh, material, first_index, index_count, transform=None, origin=(0, 0, 0), extents=(1, 1, 1)):
    ro = np.zeros((), dtype=A.RENDER_OBJECT_DTYPE)
    ro["index_count"], ro["first_index"], ro["mesh"], ro["material"] = index_count, first_index, mesh, material
    ro["origin"], ro["extents"] = origin, extents
    ro["sphere_radius"] = float(np.linalg.norm(np.asarray(extents, dtype=np.float64)))
    ro["transform"] = (IDENT if transform is None else transform).reshape(16)
    return ro


def objs(lst):
    return np.array(lst, dtype=A.RENDER_OBJECT_DTYPE) if lst else np.zeros(0, dtype=A.RENDER_OBJECT_DTYPE)


class Rig:
    """A context with the engine's default resources (src/vk_engine.cpp:226-283)."""

    def __init__(self, lib, w, h, color_format=A.COLOR_RGBA16F, background=(1, 1, 1, 1)):
        self.r = lib.create(w, h, color_format)
        self.r.set_option(A.OPT_COUNT_FRAGMENTS, 1)
        self.white = self.r.create_image(S.white_1x1())
        self.checker = self.r.create_image(S.checkerboard_32())
        self.nearest = self.r.create_sampler(**S.SAMPLER_NEAREST)
        self.linear = self.r.create_sampler(**S.SAMPLER_LINEAR)
        self.trilinear = self.r.create_sampler(**S.SAMPLER_TRILINEAR)
        self.background = background

    def material(self, color=(1, 1, 1, 1), image=None, sampler=None, transparent=False):
        return self.r.write_material(A.PASS_TRANSPARENT if transparent else A.PASS_MAIN_COLOR, color,
                                     image if image is not None else self.white,
                                     sampler if sampler is not None else self.linear)

    def draw(self, scene, opaque, transparent=()):
        self.r.clear_color(self.background)
        st = self.r.draw_geometry(scene, objs(list(opaque)), objs(list(transparent)))
        return st

    def finish(self):
        out = T._finish(self.r)
        self.r.close()
        return out


QUAD_IDX = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)


def clip_quad(x0, y0, x1, y1, z, normal=(0, 1, 0), color=(1, 1, 1, 1), uv_scale=1.0):
    """Axis-aligned quad in clip space (w = 1), two triangles sharing the (x0,y0)-(x1,y1) diagonal."""
    pos = [(x0, y0, z), (x1, y0, z), (x1, y1, z), (x0, y1, z)]
    uv = [(0, 0), (uv_scale, 0), (uv_scale, uv_scale), (0, uv_scale)]
    return make_vertices(pos, [normal] * 4, uv, [color] * 4)


# ----------------------------------------------------------------------------------------------
def shading_constants(lib, normal, size=32):
    """Full-screen quad, 1x1 white texture, vertex colour 1, given normal (SURVEY.md §8c goldens)."""
    rig = Rig(lib, size, size)
    mesh = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 1, 0.5, normal=normal))
    mat = rig.material()
    rig.draw(identity_scene(), [render_object(mesh, mat, 0, 6)])
    return rig.finish()


def shared_edge_additive(lib, size=64):
    """Two triangles sharing a diagonal, additive-blended over black: a double hit would show as 2x,
    a hole as 0 (top-left rule)."""
    rig = Rig(lib, size, size, background=(0, 0, 0, 1))
    mesh = rig.r.upload_mesh(QUAD_IDX, clip_quad(-0.83, -0.71, 0.77, 0.9, 0.5, normal=(0, 1, 0)))
    mat = rig.material(transparent=True)
    rig.draw(identity_scene(), [], [render_object(mesh, mat, 0, 6)])
    return rig.finish()


def fan_additive(lib, size=96, n=23):
    """A fan of thin triangles around an off-centre vertex, additive over black: every interior
    pixel must be hit exactly once whatever the slopes."""
    rig = Rig(lib, size, size, background=(0, 0, 0, 1))
    ang = np.linspace(0, 2 * np.pi, n + 1)
    ring = np.stack([0.93 * np.cos(ang), 0.88 * np.sin(ang), np.full_like(ang, 0.5)], axis=1)
    pos = np.concatenate([[(0.1173, -0.0631, 0.5)], ring])
    idx = np.array([[0, 1 + i, 2 + i] for i in range(n)], dtype=np.uint32).reshape(-1)
    mesh = rig.r.upload_mesh(idx, make_vertices(pos))
    mat = rig.material(transparent=True)
    rig.draw(identity_scene(), [], [render_object(mesh, mat, 0, idx.size)])
    return rig.finish()


def depth_order(lib, later_is_nearer, size=32):
    """Two overlapping quads at different depths (reversed-Z: larger = nearer), red then green."""
    rig = Rig(lib, size, size)
    za, zb = (0.25, 0.75) if later_is_nearer else (0.75, 0.25)
    ma = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 0.5, 0.5, za, color=(1, 0, 0, 1)))
    mb = rig.r.upload_mesh(QUAD_IDX, clip_quad(-0.5, -0.5, 1, 1, zb, color=(0, 1, 0, 1)))
    mat = rig.material()
    rig.draw(identity_scene(), [render_object(ma, mat, 0, 6), render_object(mb, mat, 0, 6)])
    return rig.finish()


def depth_tie(lib, size=32):
    """Coplanar identical quads, red then green with the same material: GREATER_OR_EQUAL lets the
    later draw win (src/vk_engine.cpp:1659)."""
    rig = Rig(lib, size, size)
    ma = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 1, 0.5, color=(1, 0, 0, 1)))
    mb = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 1, 0.5, color=(0, 1, 0, 1)))
    mat = rig.material()
    rig.draw(identity_scene(), [render_object(ma, mat, 0, 6), render_object(mb, mat, 0, 6)])
    return rig.finish()


def transparent_layers(lib, size=32):
    """Opaque grey quad at z=0.5 on the left half; three transparent quads: in front (z=.6) over
    everything, behind the opaque one (z=.4) and a second in-front layer.  No depth write."""
    rig = Rig(lib, size, size, background=(0.25, 0.25, 0.25, 1))
    op = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 0, 1, 0.5, color=(0.5, 0.5, 0.5, 1)))
    t1 = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 0, 0.6, color=(0.3, 0.0, 0.0, 1)))
    t2 = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 1, 0.4, color=(0.0, 0.2, 0.0, 1)))
    t3 = rig.r.upload_mesh(QUAD_IDX, clip_quad(-0.5, -1, 0.5, 1, 0.7, color=(0.0, 0.0, 0.1, 1)))
    mo, mt = rig.material(), rig.material(transparent=True)
    rig.draw(identity_scene(ambient=0.0, sun=(0, 1, 0, 1)),
             [render_object(op, mo, 0, 6)],
             [render_object(t1, mt, 0, 6), render_object(t2, mt, 0, 6), render_object(t3, mt, 0, 6)])
    return rig.finish()


def textured_plane(lib, sampler_kind, size=64, tiles=23.5, mip=True):
    """A quad with the 32x32 checker repeated `tiles` times: heavy minification."""
    rig = Rig(lib, size, size)
    img = rig.r.create_image(S.checkerboard_32(), mipmapped=mip)
    smp = {"nearest": rig.nearest, "linear": rig.linear, "trilinear": rig.trilinear}[sampler_kind]
    mesh = rig.r.upload_mesh(QUAD_IDX, clip_quad(-1, -1, 1, 1, 0.5, uv_scale=tiles))
    mat = rig.material(image=img, sampler=smp)
    rig.draw(identity_scene(ambient=0.0, sun=(0, 1, 0, 1)), [render_object(mesh, mat, 0, 6)])
    return rig.finish()


def perspective_floor(lib, w=96, h=64, sampler_kind="trilinear"):
    """A ground plane under a real camera: perspective-correct uv, anisotropic minification, the far
    end clipped by z>=0 and the near end by z<=w (it extends behind the camera)."""
    rig = Rig(lib, w, h)
    img = rig.r.create_image(S.checkerboard_32(), mipmapped=True)
    smp = {"nearest": rig.nearest, "linear": rig.linear, "trilinear": rig.trilinear}[sampler_kind]
    L = 3.0e4
    pos = [(-L, 0, -L), (L, 0, -L), (L, 0, L), (-L, 0, L)]
    uv = [(0, 0), (L, 0), (L, L), (0, L)]
    mesh = rig.r.upload_mesh(QUAD_IDX, make_vertices(pos, [(0, 1, 0)] * 4, uv))
    mat = rig.material(image=img, sampler=smp)
    scene = S.scene_data_struct((0.0, 1.5, 0.0), 0.0, 0.3, w, h)
    # bounds are caller data: with the true extents every corner is beyond the far plane or behind
    # the eye and is_visible (no w guard, z-range test) culls the floor although it fills the view —
    # the reference's behaviour, covered by test_is_visible.  Hand in bounds that pass.
    rig.draw(scene, [render_object(mesh, mat, 0, 6, extents=(1000, 0, 1000))])
    return rig.finish()


def near_clip_wall(lib, w=80, h=60):
    """A wall the camera nearly touches and that crosses the near plane and the guard band."""
    rig = Rig(lib, w, h)
    img = rig.r.create_image(S.checkerboard_32(), mipmapped=True)
    pos = [(-50, -40, -3.0), (60, -40, 1.0), (60, 45, 1.0), (-50, 45, -3.0)]
    uv = [(0, 0), (9, 0), (9, 7), (0, 7)]
    mesh = rig.r.upload_mesh(QUAD_IDX, make_vertices(pos, [(0.3, 0.5, 1)] * 4, uv))
    mat = rig.material(image=img, sampler=rig.trilinear)
    scene = S.scene_data_struct((0.0, 0.0, 0.0), 0.1, -0.2, w, h)
    # true bounds would be culled by is_visible's missing w<=0 guard (box straddles the eye plane)
    rig.draw(scene, [render_object(mesh, mat, 0, 6, origin=(0, 0, -5), extents=(1, 1, 1))])
    return rig.finish()


def depth_plane(lib, distance, size=16):
    """Camera at the origin looking down -z at a big quad `distance` away."""
    rig = Rig(lib, size, size)
    e = distance * 4.0
    pos = [(-e, -e, -distance), (e, -e, -distance), (e, e, -distance), (-e, e, -distance)]
    mesh = rig.r.upload_mesh(QUAD_IDX, make_vertices(pos))
    mat = rig.material()
    scene = S.scene_data_struct((0.0, 0.0, 0.0), 0.0, 0.0, size, size)
    rig.draw(scene, [render_object(mesh, mat, 0, 6, extents=(e, e, 0), origin=(0, 0, -distance))])
    return rig.finish()


def random_soup(lib, w=160, h=96, n_tris=600, seed=7, transparent_every=5, color_format=A.COLOR_RGBA16F,
                scissor=None, big=3):
    """Seeded triangle soup in clip space with w != 1, all sizes incl. sub-pixel and screen-filling,
    some beyond the guard band, mixed opaque/transparent, textured."""
    rng = np.random.default_rng(seed)
    rig = Rig(lib, w, h, color_format)
    img = rig.r.create_image(S.make_texture(np.random.default_rng(seed + 1), 64, 0), mipmapped=True)
    n = n_tris
    centre = rng.uniform(-1.1, 1.1, (n, 1, 2))
    scale = 10.0 ** rng.uniform(-2.5, -0.3, (n, 1, 1))
    scale[:big] = 3.0
    scale[big:2 * big] = 4.0e3  # far outside the guard band: goes through the clipper
    xy = centre + rng.normal(0, 1, (n, 3, 2)) * scale
    wv = rng.uniform(0.6, 2.5, (n, 3, 1))
    wv[-8:] = rng.uniform(-0.5, 1.0, (8, 3, 1))  # some vertices behind the eye
    z = rng.uniform(0.02, 0.98, (n, 3, 1)) * wv
    pos_clip = np.concatenate([xy * wv, z, wv], axis=2).reshape(-1, 4)
    # positions must be vec3: fold w into the world matrix per triangle is not possible, so use a
    # projective viewproj instead: clip = M * (x,y,z,1) with M's last row (0,0,1,0) -> w = z_in
    pos3 = np.stack([pos_clip[:, 0], pos_clip[:, 1], pos_clip[:, 3]], axis=1)  # z_in = w
    vp = np.zeros((4, 4), dtype=f32)
    vp[0][0] = vp[1][1] = 1
    vp[2][2] = 0.45   # clip.z = 0.45*w + 0.1  (inside [0,w] for w in (0.19, ..))
    vp[3][2] = 0.1
    vp[2][3] = 1      # clip.w = z_in
    verts = make_vertices(pos3, rng.normal(0, 1, (3 * n, 3)), rng.uniform(-2, 3, (3 * n, 2)),
                          np.concatenate([rng.uniform(0.2, 1, (3 * n, 3)), np.ones((3 * n, 1))], axis=1))
    idx = np.arange(3 * n, dtype=np.uint32)
    mesh = rig.r.upload_mesh(idx, verts)
    mo = rig.material(color=(0.9, 0.8, 1.0, 1), image=img, sampler=rig.trilinear)
    mt = rig.material(color=(0.3, 0.3, 0.2, 1), image=img, sampler=rig.linear, transparent=True)
    opaque, transparent = [], []
    for t in range(n):
        ro = render_object(mesh, mt if (transparent_every and t % transparent_every == 0) else mo, 3 * t, 3,
                           extents=(1e6, 1e6, 1e6))
        (transparent if (transparent_every and t % transparent_every == 0) else opaque).append(ro)
    scene = A.scene_struct(IDENT, IDENT, vp, [0.1] * 4, (0.2, 1, 0.5, 1), (1, 1, 1, 1))
    if scissor:
        rig.r.set_scissor(*scissor)
    rig.draw(scene, opaque, transparent)
    return rig.finish()


def transparent_stack(lib, n_layers, size=48, jitter=0.0, seed=21):
    """n_layers small transparent quads stacked on the same pixels (one 32x32 tile holds 2*n_layers
    triangles): deep in-order blending.  More than 1024 layers exceeds the tile kernel's LDS sort
    capacity (2048 entries) and sorts in the global arena."""
    rng = np.random.default_rng(seed)
    rig = Rig(lib, size, size, background=(0.0, 0.0, 0.0, 1))
    verts, idx = [], []
    for i in range(n_layers):
        cx, cy = rng.uniform(-jitter, jitter, 2)
        x0, y0 = -0.62 + cx, -0.55 + cy
        c = (0.004 + 0.003 * (i % 5), 0.002 * (i % 3), 0.001 * (i % 7), 1)
        verts.append(clip_quad(x0, y0, x0 + 0.5, y0 + 0.45, 0.3 + 0.4 * (i % 11) / 11.0, color=c))
        idx.append(QUAD_IDX + 4 * i)
    back = clip_quad(-1, -1, 0.1, 1, 0.5, color=(0.2, 0.2, 0.2, 1))  # opaque: hides the layers behind z=.5 on the left
    mesh = rig.r.upload_mesh(np.concatenate(idx), np.concatenate(verts))
    mb = rig.r.upload_mesh(QUAD_IDX, back)
    mo, mt = rig.material(), rig.material(transparent=True)
    rig.draw(identity_scene(ambient=0.0, sun=(0, 1, 0, 1)), [render_object(mb, mo, 0, 6)],
             [render_object(mesh, mt, 0, 6 * n_layers)])
    return rig.finish()


def transparent_stack_clipped(lib, n_layers=120, size=48):
    """Transparent quads that cross the z = w plane: the clipper cuts every triangle into a fan of pieces with
    the parent's submission key, and tiles on the cut hold several pieces of one triangle — equal sort keys
    in one (split) tile's transparent bin."""
    rig = Rig(lib, size, size, background=(0.0, 0.0, 0.0, 1))
    verts, idx = [], []
    for i in range(n_layers):
        zl, zr = 0.15 + 0.004 * (i % 9), 1.5 + 0.01 * (i % 7)  # left edge inside, right edge beyond the plane
        y0 = -0.9 + 0.002 * (i % 5)
        c = (0.004 + 0.003 * (i % 5), 0.002 * (i % 3), 0.001 * (i % 7), 1)
        pos = [(-0.9, y0, zl), (0.9, y0, zr), (0.9, 0.9, zr), (-0.9, 0.9, zl)]
        verts.append(make_vertices(pos, [(0, 1, 0)] * 4, [(0, 0), (1, 0), (1, 1), (0, 1)], [c] * 4))
        idx.append(QUAD_IDX + 4 * i)
    mesh = rig.r.upload_mesh(np.concatenate(idx), np.concatenate(verts))
    mt = rig.material(transparent=True)
    rig.draw(identity_scene(ambient=0.0, sun=(0, 1, 0, 1)), [], [render_object(mesh, mt, 0, 6 * n_layers)])
    return rig.finish()


def ragged_draws(lib, size=48):
    """index_count not a multiple of 3, zero-length draws, an empty opaque list entry order."""
    rig = Rig(lib, size, size)
    v = np.concatenate([clip_quad(-1, -1, 0, 0, 0.5, color=(1, 0, 0, 1)), clip_quad(0, 0, 1, 1, 0.4, color=(0, 0, 1, 1))])
    idx = np.array([0, 1, 2, 0, 2, 3, 4, 5, 6, 4, 6, 7], dtype=np.uint32)
    mesh = rig.r.upload_mesh(idx, v)
    mat = rig.material()
    rig.draw(identity_scene(), [render_object(mesh, mat, 0, 5), render_object(mesh, mat, 6, 0),
                                render_object(mesh, mat, 6, 6), render_object(mesh, mat, 3, 2)])
    return rig.finish()


def empty_frame(lib, size=40):
    rig = Rig(lib, size, size)
    rig.draw(identity_scene(), [], [])
    return rig.finish()


SCENARIOS = {
    "shading_up": lambda lib: shading_constants(lib, (0, 1, 0)),
    "shading_side": lambda lib: shading_constants(lib, (1, 0, 0)),
    "shared_edge": shared_edge_additive,
    "fan": fan_additive,
    "depth_later_nearer": lambda lib: depth_order(lib, True),
    "depth_later_farther": lambda lib: depth_order(lib, False),
    "depth_tie": depth_tie,
    "transparent_layers": transparent_layers,
    "tex_nearest": lambda lib: textured_plane(lib, "nearest"),
    "tex_linear": lambda lib: textured_plane(lib, "linear"),
    "tex_trilinear": lambda lib: textured_plane(lib, "trilinear"),
    "tex_magnified": lambda lib: textured_plane(lib, "linear", tiles=0.11),
    "floor_trilinear": perspective_floor,
    "floor_nearest": lambda lib: perspective_floor(lib, sampler_kind="nearest"),
    "near_clip_wall": near_clip_wall,
    "depth_plane_85": lambda lib: depth_plane(lib, 85.0),
    "soup": random_soup,
    "soup_rgba8": lambda lib: random_soup(lib, color_format=A.COLOR_RGBA8, seed=11),
    "soup_opaque_only": lambda lib: random_soup(lib, seed=3, transparent_every=0, n_tris=900),
    "soup_scissor": lambda lib: random_soup(lib, seed=5, scissor=(13, 21, 101, 37)),
    "soup_odd_size": lambda lib: random_soup(lib, w=67, h=35, seed=9, n_tris=300),
    "transparent_stack_40": lambda lib: transparent_stack(lib, 40, jitter=0.05),
    # 2 * layers transparent triangles in one tile: from 100 layers up the tile is split into row quarters, and the
    # counting-rank sort runs with 1, 2, 4, 6 or 8 keys per thread
    "transparent_stack_100": lambda lib: transparent_stack(lib, 100, jitter=0.04),
    "transparent_stack_230": lambda lib: transparent_stack(lib, 230, jitter=0.03, seed=5),
    "transparent_stack_450": lambda lib: transparent_stack(lib, 450, jitter=0.02, seed=6),
    "transparent_stack_700": lambda lib: transparent_stack(lib, 700, jitter=0.02),
    "transparent_stack_1000": lambda lib: transparent_stack(lib, 1000, jitter=0.01, seed=8),
    # ~1000 opaque triangles per tile: split by the opaque term of the cost alone
    "soup_dense_split": lambda lib: random_soup(lib, w=96, h=64, seed=13, n_tris=6000, transparent_every=7),
    # ~9000 opaque triangles per tile: more than a quarter's row-filtered list holds in LDS (it walks the bin itself)
    "soup_very_dense_split": lambda lib: random_soup(lib, w=64, h=64, seed=17, n_tris=36000, transparent_every=9),
    "transparent_stack_1300_fallback": lambda lib: transparent_stack(lib, 1300),
    "transparent_stack_clipped": transparent_stack_clipped,
    "ragged": ragged_draws,
    "empty": empty_frame,
}
