#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz with the CPU oracle.

The reference has no golden images or fixtures and cannot run here, so these are
outputs of this repository's oracle on seeded inputs: they freeze the oracle (and therefore the
arithmetic contract) against accidental change, and give the HIP path a fixed target that does not
depend on the oracle being rebuilt.  Inputs are regenerated from code (tests/scenarios.py,
scenes.sponza_like with seed 0x53505A41); each file stores a SHA-256 of the input vertex/index/texel
bytes so a drifting generator is reported as such rather than as a renderer bug.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import scenarios as SC  # noqa: E402
import svr_testlib as T  # noqa: E402

CASES = {
    "config1_64": lambda lib: T.render_config1(lib, 64),
    "config2_160x90": lambda lib: T.render_config2(lib, 160, 90),
    "config3_128x72": lambda lib: T.render_sponza(lib, 128, 72, lod=8, tex_size=32),
    "config3_inside_column_96x54": lambda lib: T.render_sponza(lib, 96, 54, lod=8, tex_size=32,
                                                               camera=((2.5, 1.0, -5.5), 0.2, 1.0)),
    "soup_160x96": SC.random_soup,
    "floor_96x64": SC.perspective_floor,
    "near_clip_wall_80x60": SC.near_clip_wall,
    "transparent_layers_32": SC.transparent_layers,
}


def scene_fingerprint():
    sc = T.sponza_scene(8, 32)
    h = hashlib.sha256()
    for m in sc.meshes:
        h.update(m.vertices.tobytes())
        h.update(m.indices.tobytes())
    for t in sc.textures:
        h.update(t.tobytes())
    return h.hexdigest()


def mip_case(lib):
    rng = np.random.default_rng(99)
    tex = rng.integers(0, 256, (64, 64, 4), dtype=np.uint8)
    r = lib.create(8, 8)
    img = r.create_image(tex, mipmapped=True)
    levels = [r.read_image_level(img, l) for l in range(7)]
    r.close()
    return tex, levels


def main():
    ora = T.load_oracle()
    fp = scene_fingerprint()
    for name, fn in CASES.items():
        out = fn(ora)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), color=out["color"], depth=out["depth"],
                            rgba8=out["rgba8"], scene_sha256=np.array(fp))
        print(name, out["color"].shape)
    tex, levels = mip_case(ora)
    np.savez_compressed(os.path.join(HERE, "mips_64.npz"), level0=tex, **{f"level{l}": lv for l, lv in enumerate(levels) if l})
    print("mips_64", len(levels))


if __name__ == "__main__":
    main()
