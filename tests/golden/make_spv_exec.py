#!/usr/bin/env python3
"""EXECUTES the reference's committed shader programs (shaders/*.spv: what a Vulkan driver would run) on seeded
inputs with the interpreter of tests/golden/spv_machine.py and stores inputs + outputs as tests/golden/spv_exec.npz.

Container-side only: it reads /root/reference, which does not exist on the GPU box; the fixture is data (numbers in,
numbers out), no shader text or SPIR-V words are stored.  tests/test_spv_exec.py feeds the same inputs to the oracle
and (-m gpu) to the HIP library through the ABI's per-stage hooks and compares.

What is executed, per module:
  mesh.vert                    seeded vertices x (viewproj, renderMatrix, color_factors) cases  -> gl_Position, normal, colour, uv
  colored_triangle_mesh.vert   seeded vertices x render_matrix cases                             -> gl_Position, colour, uv
  colored_triangle.vert        gl_VertexIndex 0..2                                               -> gl_Position, colour
  mesh.frag                    seeded (normal, colour, uv, texel) x SceneData cases              -> outFragColor
  tex_image.frag               seeded texels                                                     -> outColor
  colored_triangle.frag        seeded colours                                                    -> outFragColor
  gradient_color.comp          every row of a few image sizes x (data1, data2) cases             -> the stored texel
  sky.comp                     every pixel of a small image (engine default push constants)      -> the stored texel
OpImageSampleImplicitLod is answered with a supplied texel (the texture unit is fixed function: C8/C9 of DESIGN.md
stay contract-only); UNORM8 texels are supplied as c * fl32(1/255), the contract's conversion (C9).

Every case is run twice, in the interpreter's `strict` and `fused` modes (see spv_machine.py): the modules carry no
NoContraction, so a driver may legally produce either, and anything in between.

    python tests/golden/make_spv_exec.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from spv_machine import F32, Box, Machine, Module, Pointer  # noqa: E402

REF = "/root/reference/shaders"
OUT = os.path.join(HERE, "spv_exec.npz")
SEED = 0x53505658  # "SPVX"
MODES = ("strict", "fused")
BUILTIN_POSITION, BUILTIN_VERTEX_INDEX, BUILTIN_GLOBAL_INVOCATION_ID = 0, 42, 28
INV255 = np.float32(1.0) / np.float32(255.0)  # 0x3B808081


def f32list(a):
    return [F32(x) for x in np.asarray(a, dtype=np.float32).ravel()]


def mat_cols(m16):
    """column-major float[16] (glm memory layout) -> list of 4 columns"""
    m = np.asarray(m16, dtype=np.float32).reshape(4, 4)
    return [f32list(m[c]) for c in range(4)]


def vertex_struct(v12):
    """48-byte Vertex as 12 floats -> the module's struct {vec3 position; float uv_x; vec3 normal; float uv_y; vec4 color}"""
    v = f32list(v12)
    return [v[0:3], v[3], v[4:7], v[7], v[8:12]]


def set_struct(machine, module, storage, members):
    """fill the one block of that storage class whose member names are given"""
    for gid in module.globals_by_storage(storage):
        st = module.pointee(gid)
        names = module.member_names.get(st, {})
        if set(members) <= set(names.values()):
            val = machine.boxes[gid].value
            for k, nm in names.items():
                if nm in members:
                    val[k] = members[nm]
            return
    raise KeyError("no block with members %r" % list(members))


# ---------------------------------------------------------------- inputs
def realistic_matrices(rng):
    """(viewproj, world) pairs: the engine's projection (a14) behind fly-camera views, TRS world matrices"""
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    gm = g.load_package().glmath
    out = []
    for (w, h) in ((1920, 1080), (3840, 2160), (1700, 900), (256, 256)):
        pos = rng.uniform(-40, 40, 3)
        view = gm.camera_view(pos, rng.uniform(-0.5, 0.5), rng.uniform(-3, 3))
        sd = gm.scene_data(view, w, h)
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        world = gm.trs(rng.uniform(-30, 30, 3), q, rng.uniform(0.2, 5.0, 3))
        out.append((np.asarray(sd[2], np.float32).reshape(16), np.asarray(world, np.float32).reshape(16)))
    return out


def random_vertices(rng, n):
    v = np.zeros((n, 12), dtype=np.float32)
    v[:, 0:3] = rng.uniform(-60, 60, (n, 3))
    v[:, 3] = rng.uniform(-4, 4, n)
    nrm = rng.normal(size=(n, 3))
    v[:, 4:7] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    v[:, 7] = rng.uniform(-4, 4, n)
    v[:, 8:12] = rng.uniform(0, 1, (n, 4))
    v[: n // 8, 8:12] = 1.0          # the loader's default colour
    v[: n // 16, 4:7] = (1.0, 0.0, 0.0)  # the loader's default normal
    return v


# ---------------------------------------------------------------- runners
def run_vertex(module, mode, vertices, push_matrix, scene_viewproj=None, color_factors=None):
    """one invocation per vertex -> (clip [n,4], outputs by location {loc: [n,k]})"""
    vb = Box([[vertex_struct(v) for v in vertices]])  # VertexBuffer { Vertex vertices[]; }
    clip, outs = [], {}
    out_locs = sorted(module.location_of(g) for g in module.globals_by_storage(3) if module.location_of(g) is not None)
    for i in range(len(vertices)):
        mc = Machine(module, mode)
        mc.set_builtin(BUILTIN_VERTEX_INDEX, np.int32(i))
        mname = "renderMatrix" if scene_viewproj is not None else "render_matrix"
        set_struct(mc, module, 9, {mname: mat_cols(push_matrix), "vertexBuffer": Pointer(vb)})
        if scene_viewproj is not None:
            set_struct(mc, module, 2, {"viewproj": mat_cols(scene_viewproj)})
            set_struct(mc, module, 2, {"color_factors": f32list(color_factors)})
        mc.run()
        clip.append(mc.get_builtin_member(BUILTIN_POSITION))
        for loc in out_locs:
            outs.setdefault(loc, []).append(mc.get_location(loc))
    return np.array(clip, np.float32), {k: np.array(v, np.float32) for k, v in outs.items()}


def gen_mesh_vert(rng, doc):
    m = Module(os.path.join(REF, "mesh.vert.spv"))
    cases = realistic_matrices(rng)
    for _ in range(2):  # plus unstructured matrices: every term of every chain is live
        cases.append((rng.uniform(-2, 2, 16).astype(np.float32), rng.uniform(-2, 2, 16).astype(np.float32)))
    n = 96
    vp = np.array([c[0] for c in cases], np.float32)
    world = np.array([c[1] for c in cases], np.float32)
    cf = rng.uniform(0.5, 1.0, (len(cases), 4)).astype(np.float32)
    cf[0] = 1.0
    verts = np.array([random_vertices(rng, n) for _ in cases], np.float32)
    doc["mesh_vert.viewproj"], doc["mesh_vert.world"], doc["mesh_vert.color_factors"] = vp, world, cf
    doc["mesh_vert.vertices"] = verts
    for mode in MODES:
        clips, varys = [], []
        for k in range(len(cases)):
            clip, outs = run_vertex(m, mode, verts[k], world[k], vp[k], cf[k])
            clips.append(clip)
            varys.append(np.concatenate([outs[0], outs[1], outs[2]], axis=1))  # normal3, colour3, uv2 — the ABI's order
        doc["mesh_vert.clip." + mode] = np.array(clips, np.float32)
        doc["mesh_vert.varyings." + mode] = np.array(varys, np.float32)


def gen_tex_image_vert(rng, doc):
    m = Module(os.path.join(REF, "colored_triangle_mesh.vert.spv"))
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    S = g.load_package().scenes
    mats = [np.asarray(S.config2_render_matrix(1920, 1080), np.float32).reshape(16),
            rng.uniform(-2, 2, 16).astype(np.float32), rng.uniform(-2, 2, 16).astype(np.float32)]
    n = 64
    verts = np.array([random_vertices(rng, n) for _ in mats], np.float32)
    doc["tex_image_vert.render_matrix"], doc["tex_image_vert.vertices"] = np.array(mats, np.float32), verts
    for mode in MODES:
        clips, varys = [], []
        for k in range(len(mats)):
            clip, outs = run_vertex(m, mode, verts[k], mats[k])
            clips.append(clip)
            varys.append(np.concatenate([outs[0], outs[1]], axis=1))  # colour3, uv2
        doc["tex_image_vert.clip." + mode] = np.array(clips, np.float32)
        doc["tex_image_vert.varyings." + mode] = np.array(varys, np.float32)


def gen_colored_triangle(rng, doc):
    mv = Module(os.path.join(REF, "colored_triangle.vert.spv"))
    clip, col = [], []
    for i in range(3):
        mc = Machine(mv, "strict")
        mc.set_builtin(BUILTIN_VERTEX_INDEX, np.int32(i))
        mc.run()
        clip.append(mc.get_builtin_member(BUILTIN_POSITION))
        col.append(mc.get_location(0))
    doc["colored_triangle.clip"], doc["colored_triangle.color"] = np.array(clip, np.float32), np.array(col, np.float32)
    mf = Module(os.path.join(REF, "colored_triangle.frag.spv"))
    cin = rng.uniform(0, 1, (32, 3)).astype(np.float32)
    outs = []
    for c in cin:
        mc = Machine(mf, "strict")
        mc.set_location(0, f32list(c))
        mc.run()
        outs.append(mc.get_location(0))
    doc["colored_triangle.frag_in"], doc["colored_triangle.frag_out"] = cin, np.array(outs, np.float32)


def gen_mesh_frag(rng, doc):
    m = Module(os.path.join(REF, "mesh.frag.spv"))
    n_cases, n = 4, 1024
    amb = np.zeros((n_cases, 4), np.float32)
    sun_dir = np.zeros((n_cases, 4), np.float32)
    sun_col = np.zeros((n_cases, 4), np.float32)
    amb[0], sun_dir[0], sun_col[0] = 0.1, (0, 1, 0.5, 1), 1.0   # update_scene's constants, src/vk_engine.cpp:1496-1498
    for k in range(1, n_cases):
        amb[k] = rng.uniform(0, 0.4, 4)
        sun_dir[k] = rng.uniform(-1, 1, 4)
        sun_col[k] = rng.uniform(0.2, 2.5, 4)
    normal = rng.normal(size=(n_cases, n, 3))
    normal /= np.linalg.norm(normal, axis=2, keepdims=True)
    normal[:, : n // 4] *= rng.uniform(0.3, 3.0, (n_cases, n // 4, 1))   # mesh.vert does not normalise
    normal = normal.astype(np.float32)
    normal[0, 0], normal[0, 1] = (0, 1, 0), (1, 0, 0)                    # 1.1 -> 255 and 0.2 -> 51
    color = rng.uniform(0.0, 1.0, (n_cases, n, 3)).astype(np.float32)
    color[:, : n // 8] = 1.0
    texel = rng.integers(0, 256, (n_cases, n, 4)).astype(np.uint8)
    texel[0, 0] = texel[0, 1] = 255
    uv = rng.uniform(-2, 2, (n_cases, n, 2)).astype(np.float32)
    doc["mesh_frag.ambient_color"], doc["mesh_frag.sunlight_direction"], doc["mesh_frag.sunlight_color"] = amb, sun_dir, sun_col
    doc["mesh_frag.normal"], doc["mesh_frag.color"], doc["mesh_frag.texel"], doc["mesh_frag.uv"] = normal, color, texel, uv
    for mode in MODES:
        out = np.zeros((n_cases, n, 4), np.float32)
        for k in range(n_cases):
            for i in range(n):
                t = texel[k, i].astype(np.float32) * INV255
                mc = Machine(m, mode, sample=lambda coord, t=t: t)
                set_struct(mc, m, 2, {"ambient_color": f32list(amb[k]), "sunlight_direction": f32list(sun_dir[k]),
                                      "sunlight_color": f32list(sun_col[k])})
                mc.set_location(0, f32list(normal[k, i]))
                mc.set_location(1, f32list(color[k, i]))
                mc.set_location(2, f32list(uv[k, i]))
                mc.run()
                out[k, i] = mc.get_location(0)
        doc["mesh_frag.out." + mode] = out


def gen_tex_image_frag(rng, doc):
    m = Module(os.path.join(REF, "tex_image.frag.spv"))
    texel = rng.integers(0, 256, (256, 4)).astype(np.uint8)
    out = np.zeros((256, 4), np.float32)
    for i in range(256):
        t = texel[i].astype(np.float32) * INV255
        mc = Machine(m, "strict", sample=lambda coord, t=t: t)
        mc.set_location(0, f32list(rng.uniform(0, 1, 3)))
        mc.set_location(1, f32list(rng.uniform(0, 1, 2)))
        mc.run()
        out[i] = mc.get_location(0)
    doc["tex_image_frag.texel"], doc["tex_image_frag.out"] = texel, out


def run_compute(module, mode, size, data16, coords):
    """-> {(x, y): rgba}; invocations outside the image must store nothing"""
    stores = {}

    def write(coord, value):
        stores[tuple(coord)] = [float(v) for v in value]

    d = np.asarray(data16, np.float32).reshape(4, 4)
    for (x, y) in coords:
        mc = Machine(module, mode, image_size=size, image_write=write)
        mc.set_builtin(BUILTIN_GLOBAL_INVOCATION_ID, [np.uint32(x), np.uint32(y), np.uint32(0)])
        set_struct(mc, module, 9, {"data1": f32list(d[0]), "data2": f32list(d[1]), "data3": f32list(d[2]), "data4": f32list(d[3])})
        mc.run()
    return stores


def gen_gradient(rng, doc):
    m = Module(os.path.join(REF, "gradient_color.comp.spv"))
    assert m.local_size == (16, 16, 1)
    sizes = [(8, 48), (4, 1080), (4, 2160), (6, 97)]
    datas = [np.array((1, 1, 1, 1) * 2 + (0,) * 8, np.float32)]          # src/vk_engine.cpp:981-982
    datas.append(np.array((1, 0, 0, 1, 0, 0, 1, 1) + (0,) * 8, np.float32))  # vkguide's red -> blue
    for _ in range(2):
        datas.append(np.concatenate([rng.uniform(0, 1.5, 8), np.zeros(8)]).astype(np.float32))
    doc["gradient.sizes"], doc["gradient.data"] = np.array(sizes, np.int32), np.array(datas, np.float32)
    for mode in MODES:
        for si, (w, h) in enumerate(sizes):
            rows = np.zeros((len(datas), h, 4), np.float32)
            for di, d in enumerate(datas):
                st = run_compute(m, mode, (w, h), d, [(0, y) for y in range(h)] + [(w - 1, h // 2), (w, 0), (0, h), (w + 5, h + 5)])
                assert (w, 0) not in st and (0, h) not in st and (w + 5, h + 5) not in st, "store outside the image"
                assert st[(w - 1, h // 2)] == st[(0, h // 2)], "the gradient depends on x"
                for y in range(h):
                    rows[di, y] = st[(0, y)]
            doc["gradient.rows.%d.%s" % (si, mode)] = rows


def gen_sky(rng, doc):
    m = Module(os.path.join(REF, "sky.comp.spv"))
    w, h = 48, 32
    data = np.array((0.1, 0.2, 0.4, 0.97) + (0,) * 12, np.float32)      # src/vk_engine.cpp:988
    st = run_compute(m, "strict", (w, h), data, [(x, y) for y in range(h) for x in range(w)] + [(w, 0), (0, h)])
    assert (w, 0) not in st and (0, h) not in st
    img = np.zeros((h, w, 4), np.float32)
    for (x, y), v in st.items():
        img[y, x] = v
    doc["sky.size"], doc["sky.data"], doc["sky.image.strict"] = np.array((w, h), np.int32), data, img


def main():
    if not os.path.isdir(REF):
        print("reference tree not present: nothing to do", file=sys.stderr)
        return 1
    rng = np.random.default_rng(SEED)
    doc = {}
    for fn in (gen_mesh_vert, gen_tex_image_vert, gen_colored_triangle, gen_mesh_frag, gen_tex_image_frag, gen_gradient, gen_sky):
        fn(rng, doc)
        print(fn.__name__, "done", flush=True)
    np.savez_compressed(OUT, **doc)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
