#!/usr/bin/env python3
"""Records the facts the oracle's restatement of the shaders relies on, read from the reference's COMMITTED
SPIR-V (shaders/*.spv, what a Vulkan driver would actually run), into tests/golden/spv_facts.json.

Runs in the container only (it reads /root/reference); tests/test_spv_facts.py checks the recorded facts
against the constants of the arithmetic contract and, where the reference tree is
present, re-derives them.  This pins nothing numerically — the draw path stays "parity unpinned" — it stops the
restatement from drifting away from the shaders' structure: operation order of the vertex transform, which
uniform member scales the light, the clamp constant, the absence of NoContraction, the vertex layout.

    python tests/golden/make_spv_facts.py
"""
import json
import os
import struct
import sys

REF = "/root/reference/shaders"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spv_facts.json")
SHADERS = ["mesh.vert", "mesh.frag", "tex_image.frag", "colored_triangle.vert", "colored_triangle.frag", "colored_triangle_mesh.vert"]

OP = {5: "Name", 6: "MemberName", 11: "ExtInstImport", 12: "ExtInst", 43: "Constant", 22: "TypeFloat", 61: "Load", 65: "AccessChain",
      71: "Decorate", 72: "MemberDecorate", 81: "CompositeExtract", 87: "ImageSampleImplicitLod", 142: "VectorTimesScalar",
      145: "MatrixTimesVector", 146: "MatrixTimesMatrix", 148: "Dot", 129: "FAdd", 133: "FMul", 21: "TypeInt", 59: "Variable", 62: "Store"}
DEC_ARRAY_STRIDE, DEC_OFFSET, DEC_NO_CONTRACTION = 6, 35, 42
GLSL_FMAX = 40


def words(path):
    data = open(path, "rb").read()
    w = struct.unpack("<%dI" % (len(data) // 4), data)
    assert w[0] == 0x07230203, "not SPIR-V"
    return w


def literal_string(ws):
    b = b"".join(struct.pack("<I", x) for x in ws)
    return b.split(b"\0", 1)[0].decode()


def parse(path):
    w = words(path)
    i, ins = 5, []
    while i < len(w):
        n, op = w[i] >> 16, w[i] & 0xffff
        ins.append((op, w[i + 1:i + n]))
        i += n
    return {"version": (w[1] >> 16 & 0xff, w[1] >> 8 & 0xff), "ins": ins}


def facts(name):
    m = parse(os.path.join(REF, name + ".spv"))
    names, member_names, member_offsets, array_strides, consts_f, consts_i = {}, {}, {}, {}, {}, {}
    float_types, int_types, no_contraction, ext = set(), set(), False, []
    seq = []
    for op, a in m["ins"]:
        if op == 5:
            names[a[0]] = literal_string(a[1:])
        elif op == 6:
            member_names.setdefault(a[0], {})[a[1]] = literal_string(a[2:])
        elif op == 22:
            float_types.add(a[0])
        elif op == 21:
            int_types.add(a[0])
        elif op == 43:
            if a[0] in float_types:
                consts_f[a[1]] = struct.unpack("<f", struct.pack("<I", a[2]))[0]
            elif a[0] in int_types:
                consts_i[a[1]] = a[2]
        elif op == 71:
            if a[1] == DEC_ARRAY_STRIDE:
                array_strides[a[0]] = a[2]
            if a[1] == DEC_NO_CONTRACTION:
                no_contraction = True
        elif op == 72:
            if a[2] == DEC_OFFSET:
                member_offsets.setdefault(a[0], {})[a[1]] = a[3]
        elif op == 12:
            ext.append(a[3])
        if op in (145, 146, 87, 148, 12, 65, 81):
            seq.append((op, a))
    out = {"spirv_version": "%d.%d" % m["version"], "no_contraction_decoration": no_contraction,
           "float_constants": sorted(set(round(v, 9) for v in consts_f.values())),
           "glsl_std_450_instructions": sorted(set(ext)), "array_strides": sorted(array_strides.values()),
           "structs": {}}
    for sid, mem in member_names.items():
        nm = names.get(sid, str(sid))
        out["structs"][nm] = {"members": [mem[k] for k in sorted(mem)],
                              "offsets": [member_offsets.get(sid, {}).get(k) for k in sorted(mem)]}
    # order of the matrix operations: MatrixTimesMatrix whose result feeds a MatrixTimesVector
    mm = [a for op, a in seq if op == 146]
    mv = [a for op, a in seq if op == 145]
    out["matrix_times_matrix"] = len(mm)
    out["matrix_times_vector"] = len(mv)
    out["mvp_is_matrix_times_matrix_then_vector"] = bool(mm) and any(v[2] == mm[0][1] for v in mv)
    # FMax(x, const): the clamp of the Lambert term
    fmax = [a for op, a in seq if op == 12 and a[3] == GLSL_FMAX]
    out["fmax_constants"] = sorted(round(consts_f[x], 9) for a in fmax for x in a[4:] if x in consts_f)
    # which member / component of which uniform block is read: AccessChain(base, int consts...) [+ CompositeExtract]
    chains = []
    for op, a in seq:
        if op == 65:
            idx = [consts_i.get(x) for x in a[3:]]
            if all(v is not None for v in idx):
                chains.append({"base": names.get(a[2], str(a[2])), "indices": idx})
    out["access_chains"] = chains
    out["image_sample_implicit_lod"] = sum(1 for op, _ in seq if op == 87)
    out["dot_products"] = sum(1 for op, _ in seq if op == 148)
    return out


def main():
    if not os.path.isdir(REF):
        print("reference tree not present: nothing to do", file=sys.stderr)
        return 1
    doc = {"_about": "facts read from the reference's committed SPIR-V by tests/golden/make_spv_facts.py (no shader text is stored)"}
    for s in SHADERS:
        doc[s] = facts(s)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    print("wrote", OUT)
    return 0


if __name__ == "__main__":
    sys.exit(main())
