"""Synthetic inputs for the BASELINE.json configs (no glTF asset ships with the reference:
`assets/` is git-ignored).  Everything here produces *inputs* of the draw path in the
exact layouts the reference's loader produces them (src/vk_loader.cpp:286-380): one shared
vertex+index buffer per mesh, indices rebased per primitive, one GeoSurface per primitive with the
loader's bounds rule, one RenderObject per surface per frame (src/vk_engine.cpp:1716-1736).
"""
import math

import numpy as np

from . import abi, glmath

f32 = np.float32
SPONZA_SEED = 0x53505A41


# ----------------------------------------------------------------------------- containers
class Surface:
    def __init__(self, start_index, count, material, origin, radius, extents):
        self.start_index, self.count, self.material = start_index, count, material
        self.origin, self.radius, self.extents = origin, radius, extents


class MeshAsset:
    def __init__(self, name):
        self.name = name
        self.indices = np.zeros(0, dtype=np.uint32)
        self.vertices = np.zeros(0, dtype=abi.VERTEX_DTYPE)
        self.surfaces = []

    def add_primitive(self, positions, normals, uvs, indices, material, colors=None):
        """One glTF primitive, packed as src/vk_loader.cpp:299-377 does."""
        initial_vtx = self.vertices.size
        n = positions.shape[0]
        v = np.zeros(n, dtype=abi.VERTEX_DTYPE)
        v["position"] = positions.astype(f32)
        v["normal"] = normals.astype(f32) if normals is not None else np.array([1, 0, 0], dtype=f32)
        v["color"] = colors.astype(f32) if colors is not None else f32(1)
        if uvs is not None:
            v["uv_x"] = uvs[:, 0].astype(f32)
            v["uv_y"] = uvs[:, 1].astype(f32)
        start = self.indices.size
        self.indices = np.concatenate([self.indices, (indices.astype(np.uint32) + np.uint32(initial_vtx))])
        self.vertices = np.concatenate([self.vertices, v])
        # bounds: min/max start at this primitive's first vertex but the loop runs over ALL vertices
        # accumulated so far in the mesh (src/vk_loader.cpp:366-371)
        allpos = self.vertices["position"]
        first = allpos[initial_vtx]
        mn = np.minimum(first, allpos.min(axis=0))
        mx = np.maximum(first, allpos.max(axis=0))
        origin = (mx + mn) / f32(2)
        extents = (mx - mn) / f32(2)
        e2 = (extents * extents).astype(f32)  # glm::length = sqrt(dot): fp32 products, (x + y) + z, fp32 sqrt
        radius = f32(np.sqrt(f32(f32(e2[0] + e2[1]) + e2[2])))
        surf = Surface(start, indices.size, material, origin.astype(f32), radius, extents.astype(f32))
        surf.first_vertex, surf.n_vertices = initial_vtx, n  # the primitive's own slice (gltf_io.write_glb)
        self.surfaces.append(surf)


class Scene:
    """A LoadedGLTF-shaped bag: textures, samplers, materials, meshes, mesh nodes."""

    def __init__(self):
        self.textures = []      # uint8 [h,w,4]
        self.texture_mips = []  # bool
        self.samplers = []      # dict(mag,minf,mip,min_lod,max_lod)
        self.materials = []     # dict(pass_type, color_factors, texture, sampler)
        self.meshes = []        # MeshAsset
        self.nodes = []         # (mesh index, world_transform 4x4)

    # upload through the C ABI, like load_gltf_meshes does through the engine
    def upload(self, r):
        h = {"images": [], "samplers": [], "materials": [], "meshes": []}
        for tex, mips in zip(self.textures, self.texture_mips):
            h["images"].append(r.create_image(tex, mipmapped=mips))
        for s in self.samplers:
            h["samplers"].append(r.create_sampler(**s))
        for m in self.materials:
            h["materials"].append(r.write_material(m["pass_type"], m["color_factors"],
                                                   h["images"][m["texture"]], h["samplers"][m["sampler"]]))
        for mesh in self.meshes:
            h["meshes"].append(r.upload_mesh(mesh.indices, mesh.vertices))
        return h

    def render_objects(self, handles, top_matrix=None, instance_transforms=None):
        """LoadedGLTF::Draw -> (opaque, transparent) RenderObject arrays.

        node_matrix = world_transform * top_matrix (src/vk_engine.cpp:1717).  With
        instance_transforms (list of 4x4) the objects are emitted once per instance with
        transform = instance * world_transform (config 5's instancing).
        """
        top = glmath.identity() if top_matrix is None else top_matrix
        opaque, transparent = [], []
        insts = [None] if instance_transforms is None else instance_transforms
        for inst in insts:
            for mesh_idx, world in self.nodes:
                node_matrix = glmath.matmul(world, top) if inst is None else glmath.matmul(inst, world)
                mesh = self.meshes[mesh_idx]
                for s in mesh.surfaces:
                    ro = np.zeros((), dtype=abi.RENDER_OBJECT_DTYPE)
                    ro["index_count"], ro["first_index"] = s.count, s.start_index
                    ro["mesh"] = handles["meshes"][mesh_idx]
                    ro["material"] = handles["materials"][s.material]
                    ro["origin"], ro["sphere_radius"], ro["extents"] = s.origin, s.radius, s.extents
                    ro["transform"] = node_matrix.reshape(16)
                    if self.materials[s.material]["pass_type"] == abi.PASS_TRANSPARENT:
                        transparent.append(ro)
                    else:
                        opaque.append(ro)
        mk = lambda lst: np.array(lst, dtype=abi.RENDER_OBJECT_DTYPE) if lst else np.zeros(0, dtype=abi.RENDER_OBJECT_DTYPE)
        return mk(opaque), mk(transparent)

    def counts(self):
        tris = sum(int(m.indices.size) // 3 for m in self.meshes)
        verts = sum(int(m.vertices.size) for m in self.meshes)
        surfs = sum(len(m.surfaces) for m in self.meshes)
        return {"triangles": tris, "vertices": verts, "meshes": len(self.meshes), "surfaces": surfs,
                "materials": len(self.materials), "nodes": len(self.nodes)}


# ----------------------------------------------------------------------------- defaults (a20)
def checkerboard_32():
    """_error_checkerboard_image, src/vk_engine.cpp:241-250: magenta FF 00 FF FF / black 00 00 00 FF."""
    img = np.zeros((32, 32, 4), dtype=np.uint8)
    y, x = np.mgrid[0:32, 0:32]
    mag = ((x % 2) ^ (y % 2)).astype(bool)
    img[..., 3] = 255
    img[mag, 0] = 255
    img[mag, 2] = 255
    return img


def white_1x1():
    return np.full((1, 1, 4), 255, dtype=np.uint8)


SAMPLER_NEAREST = dict(mag=abi.FILTER_NEAREST, minf=abi.FILTER_NEAREST, mip=abi.MIPMAP_NEAREST, min_lod=0.0, max_lod=0.0)
SAMPLER_LINEAR = dict(mag=abi.FILTER_LINEAR, minf=abi.FILTER_LINEAR, mip=abi.MIPMAP_NEAREST, min_lod=0.0, max_lod=0.0)
SAMPLER_TRILINEAR = dict(mag=abi.FILTER_LINEAR, minf=abi.FILTER_LINEAR, mip=abi.MIPMAP_LINEAR, min_lod=0.0,
                         max_lod=abi.LOD_CLAMP_NONE)


# ----------------------------------------------------------------------------- config 2: cube
def cube_mesh():
    """Unit cube, 24 vertices / 36 indices, per-face uv in [0,1]^2, axial normals, colour 1."""
    faces = [  # normal, u axis, v axis
        ((0, 0, 1), (1, 0, 0), (0, 1, 0)), ((0, 0, -1), (-1, 0, 0), (0, 1, 0)),
        ((1, 0, 0), (0, 0, -1), (0, 1, 0)), ((-1, 0, 0), (0, 0, 1), (0, 1, 0)),
        ((0, 1, 0), (1, 0, 0), (0, 0, -1)), ((0, -1, 0), (1, 0, 0), (0, 0, 1)),
    ]
    pos, nrm, uv, idx = [], [], [], []
    for n, ua, va in faces:
        n, ua, va = np.array(n, f32), np.array(ua, f32), np.array(va, f32)
        base = len(pos)
        for (cu, cv) in ((0, 0), (1, 0), (1, 1), (0, 1)):
            pos.append(n * f32(0.5) + ua * f32(cu - 0.5) + va * f32(cv - 0.5))
            nrm.append(n)
            uv.append((cu, cv))
        idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    m = MeshAsset("Cube")
    m.add_primitive(np.array(pos, f32), np.array(nrm, f32), np.array(uv, f32), np.array(idx, np.uint32), 0)
    return m


def config2_render_matrix(width=1920, height=1080):
    """P*V*M of SURVEY.md §8d config 2."""
    proj = glmath.perspective_rh_zo(glmath.radians(70.0), f32(width) / f32(height), 10000.0, 0.1)
    proj[1][1] *= f32(-1)
    view = glmath.translate(glmath.identity(), (0, 0, -5))
    model = glmath.matmul(glmath.rotate(glmath.identity(), glmath.radians(30.0), (0, 1, 0)),
                          glmath.rotate(glmath.identity(), glmath.radians(20.0), (1, 0, 0)))
    return glmath.matmul(glmath.matmul(proj, view), model)


# ----------------------------------------------------------------------------- textures
def _value_noise(rng, size, cells):
    g = rng.random((cells, cells)).astype(f32)
    x = np.arange(size, dtype=f32) * f32(cells / size)
    i0 = np.floor(x).astype(np.int64)
    f = x - i0
    f = f * f * (3 - 2 * f)
    i0 %= cells
    i1 = (i0 + 1) % cells
    a = g[np.ix_(i0, i0)]
    b = g[np.ix_(i0, i1)]
    c = g[np.ix_(i1, i0)]
    d = g[np.ix_(i1, i1)]
    fx, fy = f[None, :], f[:, None]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_texture(rng, size, kind):
    """Seeded tileable RGBA8 texture: multi-octave value noise modulated by a brick/stripe/checker pattern."""
    size = int(size)
    n = np.zeros((size, size), dtype=f32)
    amp, tot, cells = 1.0, 0.0, 4
    while cells <= max(4, size // 2) and amp > 0.05:
        n += f32(amp) * _value_noise(rng, size, min(cells, size))
        tot += amp
        amp *= 0.5
        cells *= 2
    n /= f32(tot)
    y, x = np.mgrid[0:size, 0:size].astype(f32) / f32(size)
    if kind % 3 == 0:    # bricks
        rows = 8
        row = np.floor(y * rows)
        xb = (x * 4 + 0.5 * (row % 2)) % 1.0
        yb = (y * rows) % 1.0
        mortar = ((xb < 0.06) | (yb < 0.1)).astype(f32)
        pat = 1.0 - 0.7 * mortar
    elif kind % 3 == 1:  # stripes
        pat = 0.6 + 0.4 * np.sign(np.sin(2 * np.pi * 6 * (x + 0.3 * n)))
    else:                # checker
        pat = 0.55 + 0.45 * (((np.floor(x * 8) + np.floor(y * 8)) % 2) * 2 - 1)
    base = rng.random(3).astype(f32) * 0.6 + 0.4
    img = np.empty((size, size, 4), dtype=np.uint8)
    for c in range(3):
        ch = n * 0.8 + 0.2 * _value_noise(rng, size, min(8, size))
        v = np.clip(ch * pat * base[c] * 1.6, 0, 1)
        img[..., c] = np.round(v * 255).astype(np.uint8)
    img[..., 3] = 255
    return img


# ----------------------------------------------------------------------------- parametric grids
def _grid(nu, nv, fn, uv_tile=(1.0, 1.0)):
    """(nu x nv) quads over (s,t) in [0,1]^2; fn(s,t) -> positions, normals."""
    s = np.linspace(0, 1, nu + 1, dtype=np.float64)
    t = np.linspace(0, 1, nv + 1, dtype=np.float64)
    S, T = np.meshgrid(s, t, indexing="ij")
    P, N = fn(S, T)
    pos = P.reshape(-1, 3).astype(f32)
    nrm = N.reshape(-1, 3).astype(f32)
    uv = np.stack([S.reshape(-1) * uv_tile[0], T.reshape(-1) * uv_tile[1]], axis=1).astype(f32)
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).reshape(-1)
    b = a + (nv + 1)
    idx = np.stack([a, b, b + 1, a, b + 1, a + 1], axis=1).reshape(-1).astype(np.uint32)
    return pos, nrm, uv, idx


def _add_split_grid(mesh, nu, nv, fn, materials, n_surfaces, uv_tile=(1.0, 1.0)):
    """Split the s-range into n_surfaces primitives, each with its own vertices (as a glTF export would)."""
    n_surfaces = max(1, min(n_surfaces, nu))
    edges = np.linspace(0, nu, n_surfaces + 1).round().astype(int)
    for k in range(n_surfaces):
        u0, u1 = edges[k], edges[k + 1]
        if u1 <= u0:
            continue
        s0, s1 = u0 / nu, u1 / nu
        sub = lambda S, T, s0=s0, s1=s1: fn(s0 + S * (s1 - s0), T)
        pos, nrm, uv, idx = _grid(u1 - u0, nv, sub, uv_tile)
        uv[:, 0] = (uv[:, 0] / f32(uv_tile[0]) * f32(s1 - s0) + f32(s0)) * f32(uv_tile[0])
        mesh.add_primitive(pos, nrm, uv, idx, materials[k % len(materials)])


def _plane(origin, du, dv, normal):
    o, du, dv, n = [np.array(a, dtype=np.float64) for a in (origin, du, dv, normal)]

    def fn(S, T):
        P = o + S[..., None] * du + T[..., None] * dv
        N = np.broadcast_to(n, P.shape)
        return P, N
    return fn


def _column(S, T):  # unit column: axis y in [0,1], radius 1 with entasis; S = height, T = angle
    ang = T * 2 * np.pi
    r = 1.0 - 0.15 * S + 0.12 * np.exp(-((S - 0.02) / 0.03) ** 2) + 0.18 * np.exp(-((S - 0.98) / 0.03) ** 2)
    P = np.stack([r * np.cos(ang), S, r * np.sin(ang)], axis=-1)
    N = np.stack([np.cos(ang), np.zeros_like(S), np.sin(ang)], axis=-1)
    return P, N


def _arch(S, T):  # S along the half circle, T across the thickness (z); unit radius in xy
    ang = S * np.pi
    P = np.stack([-np.cos(ang), np.sin(ang), T - 0.5], axis=-1)
    N = np.stack([np.cos(ang), -np.sin(ang), np.zeros_like(S)], axis=-1)  # faces the opening
    return P, N


def _curtain(phase):
    def fn(S, T):  # S = height (top to bottom), T = width; hangs in the xy plane, displaced in z
        z = 0.18 * np.sin(T * 2 * np.pi * 5 + phase) * (0.3 + 0.7 * S) + 0.05 * np.sin(S * 9 + phase)
        P = np.stack([T, 1.0 - S, z], axis=-1)
        dz = 0.18 * 2 * np.pi * 5 * np.cos(T * 2 * np.pi * 5 + phase) * (0.3 + 0.7 * S)
        N = np.stack([-dz, np.zeros_like(S), np.ones_like(S)], axis=-1)
        N /= np.linalg.norm(N, axis=-1, keepdims=True)
        return P, N
    return fn


def _sphere(S, T):  # S = latitude, T = longitude; pole rows give zero-area triangles on purpose
    th, ph = S * np.pi, T * 2 * np.pi
    N = np.stack([np.sin(th) * np.cos(ph), np.cos(th), np.sin(th) * np.sin(ph)], axis=-1)
    return N.copy(), N


def sponza_like(lod=1, tex_size=1024, seed=SPONZA_SEED, mipmapped=True):
    """Synthetic 'Sponza-style' atrium.  lod divides every tessellation count
    (lod=1: 262,144 triangles, 75 meshes, 338 surfaces, 25 materials of which 2 Transparent)."""
    rng = np.random.default_rng(seed)
    sc = Scene()
    for k in range(25):
        sc.textures.append(make_texture(rng, tex_size, k))
        sc.texture_mips.append(mipmapped)
    sc.samplers.append(dict(SAMPLER_TRILINEAR))
    transparent = {21, 22}
    for k in range(25):
        cf = np.concatenate([rng.random(3).astype(f32) * f32(0.5) + f32(0.5), [f32(1)]])
        sc.materials.append(dict(pass_type=abi.PASS_TRANSPARENT if k in transparent else abi.PASS_MAIN_COLOR,
                                 color_factors=cf, texture=k, sampler=0))
    d = lambda n: max(2, int(n) // int(lod))
    I = glmath.identity()
    X0, X1, ZW, HT = -5.0, 55.0, 10.0, 16.0

    def add(mesh, world=None):
        sc.meshes.append(mesh)
        sc.nodes.append((len(sc.meshes) - 1, I if world is None else world))

    m = MeshAsset("floor")
    _add_split_grid(m, d(128), d(48), _plane((X0, 0, -ZW), (X1 - X0, 0, 0), (0, 0, 2 * ZW), (0, 1, 0)), [0, 1, 2, 3], 16, (12, 4))
    add(m)
    m = MeshAsset("ceiling")
    _add_split_grid(m, d(128), d(48), _plane((X0, HT, ZW), (X1 - X0, 0, 0), (0, 0, -2 * ZW), (0, -1, 0)), [4, 5], 16, (12, 4))
    add(m)
    for side, name in ((-1, "wall_l"), (1, "wall_r")):
        m = MeshAsset(name)
        _add_split_grid(m, d(128), d(32), _plane((X0, 0, side * ZW), (X1 - X0, 0, 0), (0, HT, 0), (0, 0, -side)),
                        [6, 7, 8, 9], 32, (12, 3))
        add(m)
    for xx, nx, name in ((X0, 1, "end_back"), (X1, -1, "end_front")):
        m = MeshAsset(name)
        _add_split_grid(m, d(32), d(32), _plane((xx, 0, -ZW), (0, 0, 2 * ZW), (0, HT, 0), (nx, 0, 0)), [10], 4, (4, 3))
        add(m)
    col_x = [2.5 + 4.5 * i for i in range(12)]
    for row, zz in enumerate((-6.0, 6.0)):
        for i, xx in enumerate(col_x):
            m = MeshAsset(f"column_{row}_{i}")
            _add_split_grid(m, d(32), d(64), _column, [11, 12, 13, 14], 6, (4, 2))
            world = glmath.scale(glmath.translate(I, (xx, 0, zz)), (0.6, HT, 0.6))
            add(m, world)
    for row, zz in enumerate((-6.0, 6.0)):
        for i in range(11):
            m = MeshAsset(f"arch_{row}_{i}")
            _add_split_grid(m, d(64), d(8), _arch, [15, 16], 2, (3, 1))
            cx = 0.5 * (col_x[i] + col_x[i + 1])
            world = glmath.scale(glmath.translate(I, (cx, 6.0, zz)), (2.25, 2.0, 1.0))
            add(m, world)
    curtain_mats = [17, 18, 21, 19, 20, 22, 17, 19]
    for k in range(8):
        m = MeshAsset(f"curtain_{k}")
        _add_split_grid(m, d(64), d(64), _curtain(0.7 * k), [curtain_mats[k]], 2, (2, 2))
        row, i = k % 2, 1 + (k // 2) * 2
        zz = -6.0 if row == 0 else 6.0
        world = glmath.scale(glmath.translate(I, (col_x[i] + 0.3, 8.5, zz)), (3.9, 6.5, 1.0))
        add(m, world)
    for k in range(15):
        m = MeshAsset(f"sphere_{k}")
        _add_split_grid(m, d(32), d(32), _sphere, [23, 24], 2, (2, 1))
        r = 0.5 + 0.5 * float(rng.random())
        xx = 4.0 + 3.3 * k
        zz = (-2.5, 0.0, 2.5)[k % 3]
        q = glmath.angle_axis(0.4 * k, (0, 1, 0))
        world = glmath.matmul(glmath.translate(I, (xx, r, zz)),
                              glmath.matmul(glmath.quat_to_mat4(q), glmath.scale(I, (r, r, r))))
        add(m, world)
    return sc


def config3_camera():
    """pos=(0,2,0), yaw=90 deg, pitch=0: looks down +x, the long axis."""
    return (0.0, 2.0, 0.0), 0.0, float(glmath.radians(90.0))


def config5_instances():
    """4x4 grid of translated instances.  The pitch of the grid (30 along the
    60-long atrium, 8 across its 20) makes neighbours interpenetrate, and each instance sits a little
    higher than the one before (0.3), so no two floors or ceilings are coplanar: a ray from the camera of
    config5_camera() leaves a stack of shells — measured depth complexity 9.2 rasterised fragments per
    pixel over a fully covered frame (lod 1), 3.1 shaded (the curtains' transparent layers pile up)."""
    out = []
    for a in range(4):
        for b in range(4):
            out.append(glmath.translate(glmath.identity(), (a * 30.0, (4 * a + b) * 0.3, (b - 1.5) * 8.0)))
    return out


def config5_camera():
    """elevated (12 of the atrium's 16 m), just inside the first row, looking down the +x rows and 12
    degrees downwards: every ray crosses several instances' shells before it leaves the grid."""
    return (-3.0, 12.0, 0.0), float(glmath.radians(-12.0)), float(glmath.radians(90.0))


def scene_data_struct(position, pitch, yaw, window_w, window_h):
    view = glmath.camera_view(position, pitch, yaw)
    return abi.scene_struct(*glmath.scene_data(view, window_w, window_h))
