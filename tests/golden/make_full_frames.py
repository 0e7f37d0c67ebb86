#!/usr/bin/env python3
"""Pins the BASELINE workloads at their own sizes: renders configs[2], configs[3] and configs[4] —
each with the 25 mip-mapped 1024^2 textures bench.py uses (the reference's texture path:
src/vk_loader.cpp:218-230 -> create_image(..., mipmapped) src/vk_engine.cpp:1543-1545) — through the
CPU oracle and stores SHA-256 digests of the colour target (RGBA16F bits), the depth target, the RGBA8
read-back and of sixteen horizontal strips of each (so a mismatch names the rows), plus the counters
that do not depend on shading order, in tests/golden/full_frames.json.

The GPU box never runs the oracle at these sizes: tests/test_full_frames.py hashes the HIP frame and
compares.  Run here (container, no GPU), minutes of CPU:

    python tests/golden/make_full_frames.py [--only config3_3840x2160]
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import svr_testlib as T  # noqa: E402

OUT = os.path.join(HERE, "full_frames.json")
STRIPS = 16
TEX = 1024

# name -> (width, height, instanced)
FRAMES = {
    "config2_sponza_1920x1080": (1920, 1080, False),   # BASELINE configs[2]
    "config3_3840x2160": (3840, 2160, False),          # BASELINE configs[3]: the workload bench.py is quoted on
    "config4_x16_7680x4320": (7680, 4320, True),       # BASELINE configs[4]
}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).view(np.uint8).tobytes()).hexdigest()


def strip_rows(height, k):
    return (height * k) // STRIPS, (height * (k + 1)) // STRIPS


def digest(out, height):
    """What is compared: whole-image digests, per-strip digests, order-independent counters."""
    d = {}
    for key in ("color", "depth", "rgba8"):
        img = out[key]
        d[key] = sha(img)
        d[key + "_strips"] = [sha(img[slice(*strip_rows(height, k))]) for k in range(STRIPS)]
    st = out["stats"]
    d["counters"] = {"triangle_count": int(st.triangle_count), "drawcall_count": int(st.drawcall_count),
                     "culled_draws": int(st.culled_draws), "rasterized_fragments": int(st.rasterized_fragments),
                     "binned_triangles": int(st.binned_triangles), "covered_pixels": int((out["depth"] > 0).sum())}
    return d


def render(lib, name, threads=None):
    import __graft_entry__ as g
    S = g.load_package().scenes
    w, h, instanced = FRAMES[name]
    kw = dict(lod=1, tex_size=TEX, instrument=True, threads=threads)
    if instanced:
        kw.update(camera=S.config5_camera(), instances=S.config5_instances())
    return T.render_sponza(lib, w, h, **kw)


def scene_fingerprint():
    sc = T.sponza_scene(1, TEX)
    h = hashlib.sha256()
    for m in sc.meshes:
        h.update(m.vertices.tobytes())
        h.update(m.indices.tobytes())
    for t in sc.textures:
        h.update(t.tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    ap.add_argument("--threads", type=int, default=len(os.sched_getaffinity(0)))
    args = ap.parse_args()
    ora = T.load_oracle()
    doc = {}
    if os.path.exists(OUT):
        with open(OUT) as f:
            doc = json.load(f)
    doc["_about"] = ("SHA-256 of oracle frames at the BASELINE sizes with 25 x 1024^2 mip-mapped textures; "
                     "made by tests/golden/make_full_frames.py, compared by tests/test_full_frames.py")
    doc["scene_sha256"] = scene_fingerprint()
    doc["strips"] = STRIPS
    for name in FRAMES:
        if args.only and name != args.only:
            continue
        t0 = time.time()
        out = render(ora, name, threads=args.threads)
        w, h, _ = FRAMES[name]
        doc[name] = digest(out, h)
        c = doc[name]["counters"]
        print(f"{name}: {time.time() - t0:.1f} s, rasterised/covered = "
              f"{c['rasterized_fragments'] / max(c['covered_pixels'], 1):.2f}, covered {c['covered_pixels'] / (w * h):.3f}", flush=True)
        with open(OUT, "w") as f:
            json.dump(doc, f, indent=1, sort_keys=True)
            f.write("\n")


if __name__ == "__main__":
    main()
