"""Write a Scene (scenes.py) as a binary glTF 2.0 file (.glb) — the asset format the reference loads
with load_gltf_meshes (src/vk_loader.cpp:162-437).  Used to feed the C++ host's loader
(host/svr_gltf.cpp) with exactly the data the Python path uploads directly, so the two can be compared
bit for bit: float32 attributes verbatim, uint32 indices, matrices as float32-exact decimals, textures
as lossless PNG (stored/deflated by zlib) inside the BIN chunk.
"""
import json
import struct
import zlib

import numpy as np

from . import abi

FILTER_CODES = {(abi.FILTER_NEAREST, None): 9728, (abi.FILTER_LINEAR, None): 9729,
                (abi.FILTER_NEAREST, abi.MIPMAP_NEAREST): 9984, (abi.FILTER_LINEAR, abi.MIPMAP_NEAREST): 9985,
                (abi.FILTER_NEAREST, abi.MIPMAP_LINEAR): 9986, (abi.FILTER_LINEAR, abi.MIPMAP_LINEAR): 9987}


def png_encode(rgba, level=6):
    """uint8 [h, w, 4] -> PNG bytes (colour type 6, 8 bit, filter 0 on every row)."""
    rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
    h, w, c = rgba.shape
    assert c == 4

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xffffffff)

    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), rgba.reshape(h, w * 4)], axis=1).tobytes()
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, level)) + chunk(b"IEND", b""))


def _f(v):
    """float32 -> a Python float that prints with enough digits to round-trip the float32."""
    return float(np.format_float_positional(np.float32(v), unique=True, trim="0")) if np.isfinite(v) else 0.0


class _Bin:
    def __init__(self):
        self.data = bytearray()
        self.views = []
        self.accessors = []

    def view(self, raw, target=None):
        while len(self.data) % 4:
            self.data.append(0)
        v = {"buffer": 0, "byteOffset": len(self.data), "byteLength": len(raw)}
        if target:
            v["target"] = target
        self.data += raw
        self.views.append(v)
        return len(self.views) - 1

    def accessor(self, arr, ctype, gtype, target=None, minmax=False):
        arr = np.ascontiguousarray(arr)
        a = {"bufferView": self.view(arr.tobytes(), target), "componentType": ctype, "count": int(arr.shape[0]), "type": gtype}
        if minmax:
            a["min"] = [_f(x) for x in arr.min(axis=0)]
            a["max"] = [_f(x) for x in arr.max(axis=0)]
        self.accessors.append(a)
        return len(self.accessors) - 1


def write_glb(scene, path, node_locals=None):
    """scene: scenes.Scene.  Nodes are written flat, each with its world matrix as `matrix`, unless
    node_locals gives [(mesh or None, 4x4 local, [children])] to write a hierarchy verbatim."""
    b = _Bin()
    meshes = []
    for mesh in scene.meshes:
        prims = []
        for s in mesh.surfaces:
            v = mesh.vertices[s.first_vertex:s.first_vertex + s.n_vertices]
            idx = mesh.indices[s.start_index:s.start_index + s.count] - np.uint32(s.first_vertex)
            attrs = {"POSITION": b.accessor(v["position"], 5126, "VEC3", 34962, minmax=True),
                     "NORMAL": b.accessor(v["normal"], 5126, "VEC3", 34962),
                     "TEXCOORD_0": b.accessor(np.stack([v["uv_x"], v["uv_y"]], axis=1), 5126, "VEC2", 34962),
                     "COLOR_0": b.accessor(v["color"], 5126, "VEC4", 34962)}
            prims.append({"attributes": attrs, "indices": b.accessor(idx.astype(np.uint32), 5125, "SCALAR", 34963),
                          "material": int(s.material), "mode": 4})
        meshes.append({"name": mesh.name, "primitives": prims})
    images = [{"name": f"image{i}", "mimeType": "image/png", "bufferView": b.view(png_encode(t))}
              for i, t in enumerate(scene.textures)]
    samplers = []
    for s in scene.samplers:
        mip = s["mip"] if s.get("max_lod", 0.0) > 0.0 else None
        samplers.append({"magFilter": FILTER_CODES[(s["mag"], None)], "minFilter": FILTER_CODES[(s["minf"], mip)],
                         "wrapS": 10497, "wrapT": 10497})
    textures, tex_index, materials = [], {}, []
    for i, m in enumerate(scene.materials):
        key = (int(m["texture"]), int(m["sampler"]))
        if key not in tex_index:
            tex_index[key] = len(textures)
            textures.append({"source": key[0], "sampler": key[1]})
        mr = m.get("metal_rough", (1.0, 0.5))
        materials.append({"name": f"material{i}", "alphaMode": "BLEND" if m["pass_type"] == abi.PASS_TRANSPARENT else "OPAQUE",
                          "pbrMetallicRoughness": {"baseColorFactor": [_f(x) for x in m["color_factors"]],
                                                   "baseColorTexture": {"index": tex_index[key]},
                                                   "metallicFactor": _f(mr[0]), "roughnessFactor": _f(mr[1])}})
    if node_locals is None:
        nodes = [{"name": f"node{i}", "mesh": int(mi), "matrix": [_f(x) for x in np.asarray(world, dtype=np.float32).reshape(16)]}
                 for i, (mi, world) in enumerate(scene.nodes)]
    else:
        nodes = []
        for i, (mi, local, children) in enumerate(node_locals):
            n = {"name": f"node{i}", "matrix": [_f(x) for x in np.asarray(local, dtype=np.float32).reshape(16)]}
            if mi is not None:
                n["mesh"] = int(mi)
            if children:
                n["children"] = [int(c) for c in children]
            nodes.append(n)
    child_set = {c for n in nodes for c in n.get("children", [])}
    doc = {"asset": {"version": "2.0", "generator": "simple-vk-renderer_amd.gltf_io"},
           "scene": 0, "scenes": [{"nodes": [i for i in range(len(nodes)) if i not in child_set]}],
           "nodes": nodes, "meshes": meshes, "materials": materials, "textures": textures, "images": images,
           "samplers": samplers, "accessors": b.accessors, "bufferViews": b.views,
           "buffers": [{"byteLength": len(b.data)}]}
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    binc = bytes(b.data) + b"\0" * (-len(b.data) % 4)
    total = 12 + 8 + len(js) + 8 + len(binc)
    with open(path, "wb") as f:
        f.write(struct.pack("<4sII", b"glTF", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(binc), 0x004E4942) + binc)
    return total
