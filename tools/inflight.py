"""Developer tool: what frames in flight in SEPARATE target sets buy.  K contexts of one process (each its own
targets, streams and copy of the scene) render alternate frames; K = 1 is the product's normal form (one draw image,
passes serialised on it as the reference's barriers serialise them, src/vk_engine.cpp:1242-1262).

    python tools/inflight.py --contexts 1,2,3 --width 3840 --height 2160
"""
import argparse
import os
import sys
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--lod", type=int, default=1)
    ap.add_argument("--tex-size", type=int, default=1024)
    ap.add_argument("--instances", type=int, default=1)
    ap.add_argument("--frames", type=int, default=120)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--contexts", default="1,2,3")
    args = ap.parse_args()
    import torch
    pkg = g.load_package()
    S, A = pkg.scenes, pkg.abi
    hip = pkg.load_product_library()
    sc = S.sponza_like(lod=args.lod, tex_size=args.tex_size)
    ks = [int(k) for k in args.contexts.split(",")]
    ctxs = []
    for _ in range(max(ks)):
        r = hip.create(args.width, args.height)
        st = torch.cuda.Stream()
        r.set_stream(st.cuda_stream)
        handles = sc.upload(r)
        inst = S.config5_instances() if args.instances == 16 else None
        opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
        ctxs.append((r, st, opaque, transparent))
    pos, pitch, yaw = S.config5_camera() if args.instances == 16 else S.config3_camera()
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)

    def run(k, n):
        for i in range(n):
            r, _, o, t = ctxs[i % k]
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, o, t)
        for r, _, _, _ in ctxs[:k]:
            r.sync()

    res = {k: [] for k in ks}
    for k in ks:
        run(k, 12)
    for _ in range(args.rounds):
        for k in ks:
            t0 = time.perf_counter()
            run(k, args.frames)
            res[k].append((time.perf_counter() - t0) / args.frames * 1e3)
    base = float(np.median(res[ks[0]]))
    for k in ks:
        m = float(np.median(res[k]))
        print(f"{args.width}x{args.height} x{args.instances}  {k} context(s): {m:.4f} ms/frame (min {min(res[k]):.4f})  {m / base:.3f} of {ks[0]}")
    # every context's image is the same frame
    ref = ctxs[0][0].read_color()
    for r, _, _, _ in ctxs[1:]:
        assert np.array_equal(ref.view(np.uint8), r.read_color().view(np.uint8))
    print("images identical across contexts")


if __name__ == "__main__":
    main()
