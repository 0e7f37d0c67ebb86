"""The shaders' structure as the reference's committed SPIR-V has it (tests/golden/spv_facts.json, read out by
tests/golden/make_spv_facts.py in the container) against what the oracle and the kernels restate: operation order
of mesh.vert's transform (contract C0/C1), the uniform member that scales the light and the clamp constant of
mesh.frag (C10), the vertex and uniform layouts (include/svr.h), no NoContraction anywhere (fma allowed, which the
contract uses).  Structure only: it does not pin a single pixel — the draw path stays "parity unpinned"."""
import importlib.util
import json
import os

import numpy as np

import __graft_entry__ as g
import svr_testlib as T

pkg = g.load_package()
A = pkg.abi
PATH = os.path.join(T.GOLDEN_DIR, "spv_facts.json")


def facts():
    with open(PATH) as f:
        return json.load(f)


def test_recorded_facts_are_what_the_restatement_assumes():
    d = facts()
    vert, frag = d["mesh.vert"], d["mesh.frag"]
    # shaders/mesh.vert:29-38: gl_Position = (viewproj * renderMatrix) * vec4(position, 1): matrix x matrix FIRST
    assert vert["matrix_times_matrix"] == 1 and vert["mvp_is_matrix_times_matrix_then_vector"]
    assert vert["matrix_times_vector"] == 2  # the position, and the normal with w = 0
    assert {"base": "sceneData", "indices": [2]} in vert["access_chains"]  # viewproj is member 2
    # interleaved Vertex: ArrayStride 48, offsets 0/12/16/28/32 (src/vk_types.h:97-103) == SvrVertex
    assert 48 in vert["array_strides"]
    v = vert["structs"]["Vertex"]
    assert v["members"] == ["position", "uv_x", "normal", "uv_y", "color"] and v["offsets"] == [0, 12, 16, 28, 32]
    assert [getattr(A.SvrVertex, n).offset for n in v["members"]] == v["offsets"]
    assert vert["structs"]["constants"]["offsets"] == [0, 64]  # GPUDrawPushConstants: mat4 + buffer address
    # SceneData (std140): the same offsets as SvrSceneData
    for sh in (vert, frag):
        s = sh["structs"]["SceneData"]
        assert s["members"] == ["view", "proj", "viewproj", "ambient_color", "sunlight_direction", "sunlight_color"]
        assert s["offsets"] == [getattr(A.SvrSceneData, n).offset for n in s["members"]] == [0, 64, 128, 192, 208, 224]
    # shaders/mesh.frag:12-19: light = max(dot(N, sunlight_direction.xyz), 0.1); scaled by sunlight_COLOR.w (member 5, component 3)
    assert frag["dot_products"] == 1 and frag["glsl_std_450_instructions"] == [40]  # FMax only: no normalize, no pow, no reflect
    assert frag["fmax_constants"] == [np.float32(0.1).item().__round__(9)]
    assert {"base": "sceneData", "indices": [5, 3]} in frag["access_chains"]
    assert {"base": "sceneData", "indices": [4]} in frag["access_chains"] and {"base": "sceneData", "indices": [3]} in frag["access_chains"]
    assert not any(c["base"] == "sceneData" and c["indices"][:1] == [4] and len(c["indices"]) > 1 for c in frag["access_chains"])  # never sunlight_direction.w
    assert frag["image_sample_implicit_lod"] == 1  # one texture() call: colorTex; metalRoughTex is never sampled
    assert d["tex_image.frag"]["image_sample_implicit_lod"] == 1 and d["tex_image.frag"]["dot_products"] == 0
    # colored_triangle_mesh.vert: gl_Position = render_matrix * position, no scene matrices
    assert d["colored_triangle_mesh.vert"]["matrix_times_matrix"] == 0 and d["colored_triangle_mesh.vert"]["matrix_times_vector"] == 1
    # FMA contraction is allowed everywhere (the contract spells its fma chains out)
    assert not any(d[k]["no_contraction_decoration"] for k in d if not k.startswith("_"))


def test_oracle_shades_with_the_members_the_spirv_reads(oracle):
    """mesh.frag through the oracle with every SceneData member set apart: only ambient.xyz, sunlight_direction.xyz and
    sunlight_color.w may matter (sunlight_direction.w and sunlight_color.xyz must not)."""
    import scenarios as SC
    S = pkg.scenes

    def render(direction_w, color_rgb):
        r = oracle.create(8, 8)
        mesh = r.upload_mesh(SC.QUAD_IDX, SC.clip_quad(-1, -1, 1, 1, 0.5))
        img = r.create_image(S.white_1x1())
        mat = r.write_material(A.PASS_MAIN_COLOR, (1, 1, 1, 1), img, r.create_sampler())
        sc = SC.identity_scene()
        sc.sunlight_direction[3] = direction_w
        for k in range(3):
            sc.sunlight_color[k] = color_rgb
        r.clear_color((0, 0, 0, 1))
        r.draw_geometry(sc, SC.objs([SC.render_object(mesh, mat, 0, 6)]))
        out = r.read_color().copy()
        r.close()
        return out

    base = render(1.0, 1.0)
    assert np.array_equal(base, render(123.0, 1.0)) and np.array_equal(base, render(1.0, 0.25))


def test_facts_file_matches_the_reference_tree_when_it_is_here():
    if not os.path.isdir("/root/reference/shaders"):
        return  # the GPU box: the committed file is all there is
    spec = importlib.util.spec_from_file_location("make_spv_facts", os.path.join(T.GOLDEN_DIR, "make_spv_facts.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    d = facts()
    for s in m.SHADERS:
        assert json.loads(json.dumps(m.facts(s))) == d[s], s
