"""the C++ host's glTF loader (host/svr_gltf.cpp + svr_json.h + svr_png.h) against the
Python path.

(1) The synthetic atrium is written as a .glb (gltf_io.write_glb: float32 attributes verbatim, PNG
    textures) and loaded by svr_demo --gltf; the RenderObjects it submits (handles, index ranges, the
    loader's inflated bounds, node matrices) must equal Scene.render_objects byte for byte and its frame
    must equal the frame of the directly uploaded scene — which also proves the PNG decoder returned
    every texel.
(2) A hand-written .gltf exercises what (1) cannot: TRS nodes in a hierarchy (quirk D8), uint16 / uint8
    indices, missing NORMAL / TEXCOORD_0 / COLOR_0 / material, normalized ubyte colours, an interleaved
    bufferView with byteStride, sampler defaults, alphaMode BLEND, a data-URI buffer, palette / grey /
    16-bit / interlaced PNGs, an undecodable image (-> error checkerboard).
CPU: against the oracle library.  GPU: the same through libsvr_hip.so."""
import base64
import json
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import __graft_entry__ as g
import svr_testlib as T

pkg = g.load_package()
A, S, GL = pkg.abi, pkg.scenes, pkg.glmath
IO = __import__(pkg.__name__ + ".gltf_io", fromlist=["write_glb"])
HOST_DIR = os.path.join(g.PKG_DIR, "host")
W, H = 192, 108


def run_demo(lib_path, gltf, prefix, camera, extra=()):
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    cmd = [os.path.join(HOST_DIR, "svr_demo"), "--lib", lib_path, "--gltf", gltf, "--width", str(W), "--height", str(H),
           "--frames", "1", "--dump", prefix, "--camera", ",".join(repr(float(c)) for c in camera), *extra]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert p.returncode == 0, p.stdout
    out = {"log": p.stdout}
    out["scene"] = np.fromfile(prefix + ".scene", dtype=np.float32)
    out["opaque"] = np.fromfile(prefix + ".opaque", dtype=A.RENDER_OBJECT_DTYPE)
    out["transparent"] = np.fromfile(prefix + ".transparent", dtype=A.RENDER_OBJECT_DTYPE)
    out["color"] = np.fromfile(prefix + ".color", dtype=np.uint16).reshape(H, W, 4)
    out["depth"] = np.fromfile(prefix + ".depth", dtype=np.float32).reshape(H, W)
    out["swapchain"] = np.fromfile(prefix + ".swapchain", dtype=np.uint8)
    return out


def engine_defaults(r):
    """SvrEngine::init's resources, in its creation order (init_default_data, src/vk_engine.cpp:226-306)."""
    white = r.create_image(S.white_1x1())
    r.create_image(np.array([[[0xAA, 0xAA, 0xAA, 0xFF]]], dtype=np.uint8))
    r.create_image(np.array([[[0, 0, 0, 0xFF]]], dtype=np.uint8))
    checker = r.create_image(S.checkerboard_32())
    nearest = r.create_sampler(**S.SAMPLER_NEAREST)
    linear = r.create_sampler(**S.SAMPLER_LINEAR)
    r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), white, linear)
    return dict(white=white, checker=checker, nearest=nearest, linear=linear)


def python_frame(lib, sc, scene_floats, background=None):
    r = lib.create(W, H)
    engine_defaults(r)
    handles = sc.upload(r)
    op, tr = sc.render_objects(handles)
    scene = A.SvrSceneData.from_buffer_copy(scene_floats.tobytes())
    r.draw_background(*(background or (A.BACKGROUND_GRADIENT, A.GRADIENT_DEFAULT)))
    r.draw_geometry(scene, op, tr)
    out = T._finish(r)
    out["opaque"], out["transparent"] = op, tr
    out["swapchain"] = r.read_swapchain(W, H, A.SWAPCHAIN_B8G8R8A8)
    r.close()
    return out


def sort_like_the_engine(objs):
    """update_scene emits node by node; Scene.render_objects does too: same order, nothing to sort."""
    return objs


def check_atrium(lib, tmp_path, oracle):
    sc = S.sponza_like(lod=8, tex_size=64)
    glb = str(tmp_path / "atrium.glb")
    IO.write_glb(sc, glb)
    cam = ((30.0, 8.0, 9.7), -0.3, 3.0)
    demo = run_demo(lib.path, glb, str(tmp_path / "a"), (*cam[0], cam[1], cam[2]), extra=("--background", "1"))
    assert "75 meshes 258 surfaces 75 nodes 75 top nodes 25 materials 25 images 1 samplers" in demo["log"]
    ref = A.scene_struct(*GL.scene_data(GL.camera_view(*cam), W, H))
    assert np.allclose(demo["scene"], np.frombuffer(bytes(ref), dtype=np.float32), rtol=3e-7, atol=1e-6)
    py = python_frame(oracle, sc, demo["scene"], background=(A.BACKGROUND_SKY, A.SKY_DEFAULT))
    for name in ("opaque", "transparent"):
        assert demo[name].tobytes() == py[name].tobytes(), name
    assert len(demo["transparent"]) > 0
    T.assert_images_identical(demo["color"], py["color"], "loaded scene colour")
    T.assert_images_identical(demo["depth"], py["depth"], "loaded scene depth")
    assert np.array_equal(demo["swapchain"].reshape(H, W, 4), py["swapchain"])
    assert (demo["depth"] > 0).mean() > 0.5


# ---------------------------------------------------------------- (2) the hand-written file
def _png(w, h, ctype, depth, rows, palette=None, trns=None, interlace=0):
    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if palette is not None:
        out += chunk(b"PLTE", bytes(palette))
    if trns is not None:
        out += chunk(b"tRNS", bytes(trns))
    return out + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")


def _filtered_rows(rows, bpp):
    """Apply PNG filter types 0..4 in rotation so the decoder's unfilter paths all run."""
    out, prev = bytearray(), bytes(len(rows[0]))
    for y, row in enumerate(rows):
        ft = y % 5
        line = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            if ft == 0:
                pred = 0
            elif ft == 1:
                pred = a
            elif ft == 2:
                pred = b
            elif ft == 3:
                pred = (a + b) >> 1
            else:
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
            line[i] = (v - pred) & 0xff
        out += bytes([ft]) + line
        prev = row
    return bytes(out)


def _adam7(w, h, pixel_bytes, get):
    xo, yo, xs, ys = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    out = bytearray()
    for p in range(7):
        for y in range(yo[p], h, ys[p]):
            row = b"".join(get(x, y) for x in range(xo[p], w, xs[p]))
            if row:
                out += b"\x00" + row
    return bytes(out)


def make_test_images(rng):
    """(png bytes, expected RGBA8) for the decoder paths stb_image's 4-channel load defines."""
    imgs = []
    # 0: RGBA 8-bit, all five filters
    a = rng.integers(0, 256, (9, 7, 4), dtype=np.uint8)
    imgs.append((_png(7, 9, 6, 8, _filtered_rows([bytes(a[y].reshape(-1)) for y in range(9)], 4)), a))
    # 1: RGB 8-bit with a colour key
    b = rng.integers(0, 256, (5, 6, 3), dtype=np.uint8)
    b[2, 3] = (10, 20, 30)
    exp = np.concatenate([b, np.full((5, 6, 1), 255, np.uint8)], axis=2)
    exp[2, 3, 3] = 0
    imgs.append((_png(6, 5, 2, 8, _filtered_rows([bytes(b[y].reshape(-1)) for y in range(5)], 3), trns=(0, 10, 0, 20, 0, 30)), exp))
    # 2: palette, 4 bits per pixel, with palette alpha
    pal = rng.integers(0, 256, (16, 3), dtype=np.uint8)
    alpha = rng.integers(0, 256, 5, dtype=np.uint8)
    idx = rng.integers(0, 16, (4, 5))
    rows = b"".join(b"\x00" + bytes([(idx[y, 0] << 4) | idx[y, 1], (idx[y, 2] << 4) | idx[y, 3], idx[y, 4] << 4]) for y in range(4))
    exp = np.zeros((4, 5, 4), np.uint8)
    exp[..., :3] = pal[idx]
    exp[..., 3] = np.where(idx < 5, np.concatenate([alpha, np.full(11, 255, np.uint8)])[idx], 255)
    imgs.append((_png(5, 4, 3, 4, rows, palette=pal.reshape(-1), trns=alpha), exp))
    # 3: grey + alpha, 16 bits (high byte kept)
    ga = rng.integers(0, 65536, (3, 4, 2)).astype(">u2")
    exp = np.zeros((3, 4, 4), np.uint8)
    exp[..., 0] = exp[..., 1] = exp[..., 2] = (ga[..., 0] >> 8).astype(np.uint8)
    exp[..., 3] = (ga[..., 1] >> 8).astype(np.uint8)
    imgs.append((_png(4, 3, 4, 16, b"".join(b"\x00" + ga[y].tobytes() for y in range(3))), exp))
    # 4: grey 2 bits per pixel (scaled by 85)
    gr = rng.integers(0, 4, (2, 8))
    rows = b"".join(b"\x00" + bytes([(gr[y, 0] << 6) | (gr[y, 1] << 4) | (gr[y, 2] << 2) | gr[y, 3],
                                      (gr[y, 4] << 6) | (gr[y, 5] << 4) | (gr[y, 6] << 2) | gr[y, 7]]) for y in range(2))
    exp = np.zeros((2, 8, 4), np.uint8)
    exp[..., :3] = (gr * 85)[..., None]
    exp[..., 3] = 255
    imgs.append((_png(8, 2, 0, 2, rows), exp))
    # 5: RGBA 8-bit, Adam7 interlaced, odd extent
    c = rng.integers(0, 256, (11, 13, 4), dtype=np.uint8)
    imgs.append((_png(13, 11, 6, 8, _adam7(13, 11, 4, lambda x, y: bytes(c[y, x])), interlace=1), c))
    return imgs


def build_handwritten(tmp_path):
    """Returns (path of the .gltf, the equivalent scenes.Scene, engine-default expectations)."""
    rng = np.random.default_rng(7)
    imgs = make_test_images(rng)
    bin_ = bytearray()

    def put(raw, align=4):
        while len(bin_) % align:
            bin_.append(0)
        off = len(bin_)
        bin_.extend(raw)
        return off, len(raw)

    cube = S.cube_mesh()
    pos = cube.vertices["position"].astype(np.float32)
    nrm = cube.vertices["normal"].astype(np.float32)
    uv = np.stack([cube.vertices["uv_x"], cube.vertices["uv_y"]], axis=1).astype(np.float32)
    col8 = rng.integers(0, 256, (24, 4), dtype=np.uint8)
    views, accs = [], []

    def view(raw, stride=None):
        off, n = put(raw)
        v = {"buffer": 0, "byteOffset": off, "byteLength": n}
        if stride:
            v["byteStride"] = stride
        views.append(v)
        return len(views) - 1

    def acc(view_i, ctype, count, gtype, offset=0, normalized=False):
        a = {"bufferView": view_i, "componentType": ctype, "count": count, "type": gtype, "byteOffset": offset}
        if normalized:
            a["normalized"] = True
        accs.append(a)
        return len(accs) - 1

    # primitive A: interleaved position+normal+uv (stride 32), uint16 indices, ubyte-normalized colours
    inter = np.zeros((24, 8), np.float32)
    inter[:, 0:3], inter[:, 3:6], inter[:, 6:8] = pos, nrm, uv
    vi = view(inter.tobytes(), stride=32)
    a_pos, a_nrm, a_uv = acc(vi, 5126, 24, "VEC3", 0), acc(vi, 5126, 24, "VEC3", 12), acc(vi, 5126, 24, "VEC2", 24)
    a_col = acc(view(col8.tobytes()), 5121, 24, "VEC4", normalized=True)
    a_idx16 = acc(view(cube.indices.astype(np.uint16).tobytes()), 5123, 36, "SCALAR")
    # primitive B: positions only (shifted), uint8 indices, no material
    posb = (pos + np.float32([2.5, 0.25, 0])).astype(np.float32)
    # ... as a SPARSE accessor: three vertices of the base data are junk and come from the sparse values (u16 indices)
    base_b = posb.copy()
    base_b[[3, 7, 20]] = np.float32([[9, 9, 9], [-7, 0, 3], [1e3, 1e3, 1e3]])
    b_pos = acc(view(base_b.tobytes()), 5126, 24, "VEC3")
    accs[b_pos]["sparse"] = {"count": 3,
                             "indices": {"bufferView": view(np.uint16([3, 7, 20]).tobytes()), "componentType": 5123},
                             "values": {"bufferView": view(posb[[3, 7, 20]].tobytes())}}
    b_idx8 = acc(view(cube.indices.astype(np.uint8).tobytes()), 5121, 36, "SCALAR")
    # primitive C (second mesh): no indices at all (GenerateMeshIndices), VEC3 float colours
    tri = np.float32([[-1, 0, 0], [1, 0, 0], [0, 1.5, 0], [-1, 0, 1], [0, 1.5, 1], [1, 0, 1]])
    c_pos = acc(view(tri.tobytes()), 5126, 6, "VEC3")
    # colours: a sparse accessor WITHOUT a bufferView (zeros underneath), every element supplied (u8 indices)
    accs.append({"componentType": 5126, "count": 6, "type": "VEC3",
                 "sparse": {"count": 6, "indices": {"bufferView": view(np.uint8([0, 1, 2, 3, 4, 5]).tobytes()), "componentType": 5121},
                            "values": {"bufferView": view(np.float32([[1, 0.5, 0.25]] * 6).tobytes())}}})
    c_col = len(accs) - 1
    image_views = [view(png) for png, _ in imgs] + [view(b"not an image at all")]
    doc = {
        "asset": {"version": "2.0"},
        "buffers": [{"byteLength": 0, "uri": ""}],
        "bufferViews": views, "accessors": accs,
        "images": [{"bufferView": v, "mimeType": "image/png", "name": f"img{i}"} for i, v in enumerate(image_views)],
        "samplers": [{}, {"magFilter": 9729, "minFilter": 9985}, {"magFilter": 9728, "minFilter": 9729}],
        "textures": [{"source": 0, "sampler": 0}, {"source": 2, "sampler": 1}, {"source": 5, "sampler": 2}, {"source": 6, "sampler": 1},
                     {"source": 1, "sampler": 0}, {"source": 3, "sampler": 1}, {"source": 4, "sampler": 2}],
        "materials": [
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "baseColorFactor": [0.9, 0.8, 0.7, 1.0]}},
            {"alphaMode": "BLEND", "pbrMetallicRoughness": {"baseColorTexture": {"index": 1}}},
            {"alphaMode": "MASK", "pbrMetallicRoughness": {"baseColorTexture": {"index": 2}, "metallicFactor": 0.25}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 3}}},          # undecodable image -> checkerboard
            {"pbrMetallicRoughness": {"baseColorFactor": [0.5, 1.0, 0.5, 1.0]}},   # no texture -> white + default linear
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 4}}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 5}}},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 6}}}],
        "meshes": [
            {"name": "two", "primitives": [
                {"attributes": {"POSITION": a_pos, "NORMAL": a_nrm, "TEXCOORD_0": a_uv, "COLOR_0": a_col}, "indices": a_idx16, "material": 2},
                {"attributes": {"POSITION": b_pos}, "indices": b_idx8}]},
            {"name": "roof", "primitives": [{"attributes": {"POSITION": c_pos, "COLOR_0": c_col}, "material": 1}]}],
        "nodes": [
            {"name": "root", "translation": [0, -0.5, -9], "children": [1, 2]},
            {"name": "left", "mesh": 0, "translation": [-2.5, 0.5, -8], "rotation": [0, 0.38268343, 0, 0.92387953], "scale": [1, 1.25, 1], "children": [3]},
            {"name": "right", "mesh": 0, "matrix": [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, 2.0, 1.0, -6.0, 1]},
            {"name": "leaf", "mesh": 1, "translation": [0.5, 1.0, -5.0], "scale": [1.5, 1, 1]},
            {"name": "lonely", "mesh": 1, "translation": [3.0, -1.0, -7.0], "rotation": [0.70710677, 0, 0, 0.70710677]}],
        "scenes": [{"nodes": [0, 4]}], "scene": 0}
    # six more nodes show the other materials on the first mesh's cubes? no: materials 3..7 via extra meshes
    extra_pos = acc(view(pos.tobytes()), 5126, 24, "VEC3")
    extra_uv = acc(view(uv.tobytes()), 5126, 24, "VEC2")
    extra_idx = acc(view(cube.indices.astype(np.uint32).tobytes()), 5125, 36, "SCALAR")
    for k, m in enumerate((3, 4, 5, 6, 7, 0)):
        doc["meshes"].append({"name": f"m{m}", "primitives": [{"attributes": {"POSITION": extra_pos, "TEXCOORD_0": extra_uv},
                                                               "indices": extra_idx, "material": m}]})
        doc["nodes"].append({"mesh": 2 + k, "translation": [-4.0 + 1.6 * k, -2.0, -6.5]})
        doc["scenes"][0]["nodes"].append(5 + k)
    doc["buffers"][0] = {"byteLength": len(bin_), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(bin_)).decode()}
    path = str(tmp_path / "hand.gltf")
    with open(path, "w") as f:
        json.dump(doc, f)

    # ---- the same scene built the Python way
    sc = S.Scene()
    sc.textures = [e for _, e in imgs]
    sc.texture_mips = [True] * len(imgs)
    TRI = dict(min_lod=0.0, max_lod=1000.0)
    sc.samplers = [dict(mag=A.FILTER_NEAREST, minf=A.FILTER_NEAREST, mip=A.MIPMAP_LINEAR, **TRI),
                   dict(mag=A.FILTER_LINEAR, minf=A.FILTER_LINEAR, mip=A.MIPMAP_NEAREST, **TRI),
                   dict(mag=A.FILTER_NEAREST, minf=A.FILTER_LINEAR, mip=A.MIPMAP_LINEAR, **TRI)]
    return path, sc, doc, dict(col8=col8, pos=pos, nrm=nrm, uv=uv, posb=posb, tri=tri, cube=cube)


def python_handwritten(lib, sc, doc, d, scene_floats):
    r = lib.create(W, H)
    dflt = engine_defaults(r)
    images = [r.create_image(t, mipmapped=True) for t in sc.textures] + [dflt["checker"]]
    samplers = [r.create_sampler(**s) for s in sc.samplers]
    tex = [(images[t["source"]], samplers[t["sampler"]]) for t in doc["textures"]]
    mats, passes = [], []
    for m in doc["materials"]:
        pbr = m.get("pbrMetallicRoughness", {})
        cf = pbr.get("baseColorFactor", [1, 1, 1, 1])
        img, smp = tex[pbr["baseColorTexture"]["index"]] if "baseColorTexture" in pbr else (dflt["white"], dflt["linear"])
        p = A.PASS_TRANSPARENT if m.get("alphaMode") == "BLEND" else A.PASS_MAIN_COLOR
        mats.append(r.write_material(p, cf, img, smp, (pbr.get("metallicFactor", 1.0), pbr.get("roughnessFactor", 1.0), 0, 0)))
        passes.append(p)
    sc.materials = [dict(pass_type=p) for p in passes]
    cube = d["cube"]
    m0 = S.MeshAsset("two")
    m0.add_primitive(d["pos"], d["nrm"], d["uv"], cube.indices, 2, colors=(d["col8"].astype(np.float32) / np.float32(255)))
    m0.add_primitive(d["posb"], None, None, cube.indices, 0)
    m1 = S.MeshAsset("roof")
    m1.add_primitive(d["tri"], None, None, np.arange(6), 1, colors=np.float32([[1, 0.5, 0.25, 1]] * 6))
    sc.meshes = [m0, m1]
    for m in (3, 4, 5, 6, 7, 0):
        mm = S.MeshAsset(f"m{m}")
        mm.add_primitive(d["pos"], None, d["uv"], cube.indices, m)
        sc.meshes.append(mm)
    mesh_handles = [r.upload_mesh(m.indices, m.vertices) for m in sc.meshes]
    I = GL.identity()
    local = lambda n: (np.array(n["matrix"], np.float32).reshape(4, 4) if "matrix" in n else
                       GL.trs(n.get("translation", (0, 0, 0)), n.get("rotation", (0, 0, 0, 1)), n.get("scale", (1, 1, 1))))
    # refresh_transform hands the TOP node's parent matrix (identity) to every descendant (quirk D8):
    # world = identity * local for every node, whatever its depth
    sc.nodes = [(n["mesh"], GL.matmul(I, local(n))) for n in doc["nodes"] if "mesh" in n]
    # update_scene walks top nodes in file order, each depth first: 0 -> (1 -> 3), 2 ; then 4, then the extras
    order = [1, 3, 2, 4] + list(range(5, 11))
    by_index = {i: (n["mesh"], GL.matmul(I, local(n))) for i, n in enumerate(doc["nodes"]) if "mesh" in n}
    sc.nodes = [by_index[i] for i in order]
    op, tr = sc.render_objects({"meshes": mesh_handles, "materials": mats})
    scene = A.SvrSceneData.from_buffer_copy(scene_floats.tobytes())
    r.draw_background(A.BACKGROUND_GRADIENT, A.GRADIENT_DEFAULT)
    r.draw_geometry(scene, op, tr)
    out = T._finish(r)
    out["opaque"], out["transparent"] = op, tr
    r.close()
    return out


def check_handwritten(lib, tmp_path, oracle):
    path, sc, doc, d = build_handwritten(tmp_path)
    demo = run_demo(lib.path, path, str(tmp_path / "h"), (0.0, 0.0, 0.0, 0.0, 0.0))
    assert "gltf failed to load texture img6" in demo["log"]
    assert "8 meshes 9 surfaces 11 nodes 8 top nodes 8 materials 6 images 3 samplers" in demo["log"]
    py = python_handwritten(oracle, sc, doc, d, demo["scene"])
    assert len(demo["opaque"]) == len(py["opaque"]) and len(demo["transparent"]) == len(py["transparent"]) == 2
    for name in ("opaque", "transparent"):
        for k, (x, y) in enumerate(zip(demo[name], py[name])):
            assert x.tobytes() == y.tobytes(), (name, k, x, y)
    T.assert_images_identical(demo["color"], py["color"], "hand-written file colour")
    T.assert_images_identical(demo["depth"], py["depth"], "hand-written file depth")
    assert (demo["depth"] > 0).sum() > 500


def test_png_decoder(tmp_path):
    """svr_png.h on its own: every colour type / bit depth / filter / interlace path, a real-size image
    written by gltf_io.png_encode (dynamic Huffman blocks) and a stored-block stream; garbage is refused."""
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    exe = os.path.join(HOST_DIR, "svr_demo")
    rng = np.random.default_rng(7)
    cases = make_test_images(rng)
    big = S.make_texture(np.random.default_rng(3), 128, 1)
    cases.append((IO.png_encode(big), big))
    cases.append((IO.png_encode(big[:16, :16], level=0), big[:16, :16]))   # stored deflate blocks
    for k, (png, exp) in enumerate(cases):
        f = tmp_path / f"c{k}.png"
        f.write_bytes(png)
        r = subprocess.run([exe, "--png", str(f), "--dump", str(tmp_path / f"c{k}")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, (k, r.stdout)
        assert r.stdout.split()[:3] == ["png", str(exp.shape[1]), str(exp.shape[0])]
        got = np.fromfile(str(tmp_path / f"c{k}.rgba"), dtype=np.uint8).reshape(exp.shape)
        assert np.array_equal(got, exp), k
    for k, bad in enumerate((b"", b"\x89PNG\r\n\x1a\n", cases[0][0][:40], b"JFIF" * 10)):
        f = tmp_path / f"bad{k}.png"
        f.write_bytes(bad)
        r = subprocess.run([exe, "--png", str(f)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 1 and r.stdout.split(":")[0] in ("png", "unknown"), (k, r.stdout)


def test_gltf_loader_atrium_on_the_oracle(tmp_path, oracle):
    check_atrium(oracle, tmp_path, oracle)


def test_gltf_loader_handwritten_on_the_oracle(tmp_path, oracle):
    check_handwritten(oracle, tmp_path, oracle)


def test_gltf_loader_rejects_broken_files(tmp_path, oracle):
    subprocess.run(["make", "-s"], cwd=HOST_DIR, check=True)
    exe = os.path.join(HOST_DIR, "svr_demo")
    cases = {"notjson.gltf": b"{ this is not json", "empty.glb": b"glTF\x02\x00\x00\x00\x0c\x00\x00\x00",
             "badacc.gltf": json.dumps({"asset": {"version": "2.0"}, "meshes": [{"primitives": [{"attributes": {"POSITION": 3}}]}]}).encode(),
             # fastgltf is built for three extensions in the reference (src/vk_loader.cpp:169-173) and refuses files that require others
             "draco.gltf": json.dumps({"asset": {"version": "2.0"}, "extensionsRequired": ["KHR_draco_mesh_compression"],
                                       "extensionsUsed": ["KHR_draco_mesh_compression"]}).encode()}
    for name, body in cases.items():
        p = tmp_path / name
        p.write_bytes(body)
        r = subprocess.run([exe, "--lib", oracle.path, "--gltf", str(p)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 1 and "load_gltf_meshes" in r.stdout, (name, r.stdout)
    r = subprocess.run([exe, "--lib", oracle.path, "--gltf", str(tmp_path / "missing.glb")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "cannot read" in r.stdout


@pytest.mark.gpu
def test_gltf_loader_on_the_hip_library(tmp_path, hip, oracle):
    check_atrium(hip, tmp_path, oracle)
    check_handwritten(hip, tmp_path, oracle)


def test_python_loader_round_trip(tmp_path, oracle):
    """gltf_io.load_gltf (the Python drivers' loader) reads back what write_glb wrote: same geometry and
    textures, and the identical frame."""
    sc = S.sponza_like(lod=8, tex_size=64)
    glb = str(tmp_path / "rt.glb")
    IO.write_glb(sc, glb)
    back = IO.load_gltf(glb)
    assert back.counts() == sc.counts()
    for a, b in zip(sc.meshes, back.meshes):
        assert a.vertices.tobytes() == b.vertices.tobytes() and a.indices.tobytes() == b.indices.tobytes()
    for a, b in zip(sc.textures, back.textures):
        assert np.array_equal(a, b)
    cam = S.config3_camera()
    frames = []
    for scene_obj in (sc, back):
        r = oracle.create(W, H)
        handles = scene_obj.upload(r)
        op, tr = scene_obj.render_objects(handles)
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(S.scene_data_struct(*cam, W, H), op, tr)
        frames.append(T._finish(r))
        r.close()
    T.assert_images_identical(frames[0]["color"], frames[1]["color"], "round-trip colour")
    T.assert_images_identical(frames[0]["depth"], frames[1]["depth"], "round-trip depth")


def test_python_loader_agrees_with_the_cpp_loader(tmp_path, oracle):
    """The hand-written file through both loaders: the C++ host's frame and the Python loader's."""
    path, _sc, _doc, _d = build_handwritten(tmp_path)
    demo = run_demo(oracle.path, path, str(tmp_path / "h2"), (0.0, 0.0, 0.0, 0.0, 0.0))
    sc = IO.load_gltf(path)
    assert sc.counts()["surfaces"] == 9 and len(sc.materials) == 8
    r = oracle.create(W, H)
    engine_defaults(r)
    handles = sc.upload(r)
    op, tr = sc.render_objects(handles)
    r.draw_background(A.BACKGROUND_GRADIENT, A.GRADIENT_DEFAULT)
    r.draw_geometry(A.SvrSceneData.from_buffer_copy(demo["scene"].tobytes()), op, tr)
    out = T._finish(r)
    r.close()
    T.assert_images_identical(demo["color"], out["color"], "C++ loader vs Python loader colour")
    T.assert_images_identical(demo["depth"], out["depth"], "C++ loader vs Python loader depth")


def test_loader_decodes_jpeg_textures(tmp_path, oracle):
    """A GLB whose textures are JPEGs goes through host/svr_jpeg.h: every image loads (no checkerboard
    substitution) and the frame is close to the PNG version's (lossy textures, same geometry)."""
    sc = S.sponza_like(lod=8, tex_size=64)
    png, jpg = str(tmp_path / "p.glb"), str(tmp_path / "j.glb")
    IO.write_glb(sc, png)
    IO.write_glb(sc, jpg, image_format="JPEG")
    cam = (30.0, 8.0, 9.7, -0.3, 3.0)
    a = run_demo(oracle.path, png, str(tmp_path / "p"), cam)
    b = run_demo(oracle.path, jpg, str(tmp_path / "j"), cam)
    assert "failed to load texture" not in b["log"] and "25 images" in b["log"]
    assert np.array_equal(a["depth"], b["depth"])
    ca, cb = T.f16_bits_to_f32(a["color"]), T.f16_bits_to_f32(b["color"])
    assert 0 < np.abs(ca - cb).mean() < 0.03


@pytest.mark.parametrize("fmt", ["TGA", "BMP"])
def test_loader_decodes_the_minor_formats(tmp_path, oracle, fmt):
    """The reference gives every image to stb_image whatever its type (src/vk_loader.cpp:108, 131): a GLB whose
    textures are run-length TGA or 32-bit BMP files loads through host/svr_image.h, and — the formats being
    lossless — renders the very frame of the PNG version."""
    sc = S.sponza_like(lod=8, tex_size=64)
    png, other = str(tmp_path / "p.glb"), str(tmp_path / "o.glb")
    IO.write_glb(sc, png)
    IO.write_glb(sc, other, image_format=fmt)
    cam = (30.0, 8.0, 9.7, -0.3, 3.0)
    a = run_demo(oracle.path, png, str(tmp_path / "p"), cam)
    b = run_demo(oracle.path, other, str(tmp_path / "o"), cam)
    assert "failed to load texture" not in b["log"] and "25 images" in b["log"]
    T.assert_images_identical(a["color"], b["color"], f"{fmt} textures colour")
    T.assert_images_identical(a["depth"], b["depth"], f"{fmt} textures depth")


def test_quantized_attributes_load_like_their_float_twins(tmp_path, oracle):
    """KHR_mesh_quantization (one of the three extensions the reference's parser is built for, src/vk_loader.cpp:169):
    SHORT positions, normalized BYTE normals, normalized UNSIGNED_SHORT texture coordinates and UNSIGNED_BYTE colours.  The
    file that requires the extension renders the very frame of its twin with the same values as floats — through the C++
    loader and through the Python one."""
    pos16 = np.int16([[-300, 0, 0], [300, 0, 0], [0, 450, 0], [-300, 0, 200], [0, 450, 200], [300, 0, 200]])
    nrm8 = np.int8([[0, 0, 127], [0, 0, 127], [0, 0, 127], [0, 127, 0], [-128, 0, 0], [0, -127, 0]])
    uv16 = np.uint16([[0, 0], [65535, 0], [32768, 65535], [0, 13107], [13107, 13107], [52428, 0]])
    col8 = np.uint8([[255, 0, 0, 255], [0, 255, 0, 255], [0, 0, 255, 255], [255, 255, 0, 128], [0, 255, 255, 255], [51, 102, 204, 255]])
    as_float = {"POSITION": pos16.astype(np.float32), "NORMAL": np.maximum(nrm8.astype(np.float32) / np.float32(127.0), np.float32(-1.0)),
                "TEXCOORD_0": uv16.astype(np.float32) / np.float32(65535.0), "COLOR_0": col8.astype(np.float32) / np.float32(255.0)}

    def pad(raw):  # elements of quantized accessors must start on 4-byte boundaries: pad every element row
        return raw

    def write(name, quantized):
        bin_, views, accs = bytearray(), [], []

        def add(raw, ctype, gtype, count, normalized=False, stride=None):
            while len(bin_) % 4:
                bin_.append(0)
            v = {"buffer": 0, "byteOffset": len(bin_), "byteLength": len(raw)}
            if stride:
                v["byteStride"] = stride
            bin_.extend(raw)
            views.append(v)
            a = {"bufferView": len(views) - 1, "componentType": ctype, "count": count, "type": gtype}
            if normalized:
                a["normalized"] = True
            accs.append(a)
            return len(accs) - 1

        if quantized:
            p4 = np.zeros((6, 4), np.int16); p4[:, :3] = pos16   # 6-byte elements padded to a stride of 8
            n4 = np.zeros((6, 4), np.int8); n4[:, :3] = nrm8     # 3-byte elements padded to a stride of 4
            attrs = {"POSITION": add(p4.tobytes(), 5122, "VEC3", 6, stride=8), "NORMAL": add(n4.tobytes(), 5120, "VEC3", 6, True, stride=4),
                     "TEXCOORD_0": add(uv16.tobytes(), 5123, "VEC2", 6, True), "COLOR_0": add(col8.tobytes(), 5121, "VEC4", 6, True)}
        else:
            attrs = {k: add(v.astype(np.float32).tobytes(), 5126, {2: "VEC2", 3: "VEC3", 4: "VEC4"}[v.shape[1]], 6) for k, v in as_float.items()}
        doc = {"asset": {"version": "2.0"}, "bufferViews": views, "accessors": accs,
               "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 1.0]}}],
               "meshes": [{"primitives": [{"attributes": attrs, "material": 0}]}],
               "nodes": [{"mesh": 0, "translation": [0, -1.5, -9], "scale": [0.01, 0.01, 0.01]}], "scenes": [{"nodes": [0]}], "scene": 0,
               "buffers": [{"byteLength": len(bin_), "uri": "data:application/octet-stream;base64," + base64.b64encode(bytes(bin_)).decode()}]}
        if quantized:
            doc["extensionsUsed"] = doc["extensionsRequired"] = ["KHR_mesh_quantization"]
        path = str(tmp_path / name)
        with open(path, "w") as f:
            json.dump(doc, f)
        return path

    cam = (0.0, 0.0, 0.0, 0.0, 0.0)
    q, f = write("quant.gltf", True), write("float.gltf", False)
    a, b = run_demo(oracle.path, q, str(tmp_path / "q"), cam), run_demo(oracle.path, f, str(tmp_path / "f"), cam)
    assert (a["depth"] > 0).sum() > 500, "the mesh should be in view"
    T.assert_images_identical(a["color"], b["color"], "quantized vs float colour")
    T.assert_images_identical(a["depth"], b["depth"], "quantized vs float depth")
    sq, sf = IO.load_gltf(q), IO.load_gltf(f)
    assert np.array_equal(sq.meshes[0].vertices, sf.meshes[0].vertices)
