"""Developer tool: interleaved wall-clock A/B of two builds of libsvr_hip.so in ONE process on ONE
box (box-to-box variance on the pool is 15-25 %, larger than most kernel changes).

    python tools/ab_libs.py build_ab/libsvr_hip_prev.so simple-vk-renderer_amd/csrc/libsvr_hip.so:2
(path[:SVR_OPT_TUNING mask] per entry)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--instances", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--frames", type=int, default=40)
    ap.add_argument("--tuning", type=int, default=0)
    ap.add_argument("--stages", action="store_true")
    ap.add_argument("--stream", action="store_true", help="render on a stream of its own (torch.cuda.Stream) instead of the null stream")
    ap.add_argument("--band", default="", help="y0,rows: render only these rows (scissor), as one rank of a sharded frame does")
    ap.add_argument("--interleave", default="", help="stride,offset: render only the tile rows t with t %% stride == offset (svr_set_row_interleave)")
    args = ap.parse_args()
    pkg = g.load_package()
    import torch  # noqa: F401  (one HIP runtime per process: torch's is loaded first)
    S, A = pkg.scenes, pkg.abi
    sc = S.sponza_like(lod=1, tex_size=1024)
    pos, pitch, yaw = S.config5_camera() if args.instances == 16 else S.config3_camera()
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)
    inst = S.config5_instances() if args.instances == 16 else None
    ctx, keep = [], []
    for spec in args.libs:  # path[:tuning]
        path, _, tun = spec.partition(":")
        lib = A.SvrLib(os.path.abspath(path))
        r = lib.create(args.width, args.height)
        handles = sc.upload(r)
        opaque, transparent = sc.render_objects(handles, instance_transforms=inst)
        r.set_option(A.OPT_TUNING, int(tun) if tun else args.tuning)
        if args.stream:
            st = torch.cuda.Stream()
            keep.append(st)
            r.set_stream(st.cuda_stream)
        if args.band:
            y0, rows = (int(v) for v in args.band.split(","))
            r.set_scissor(0, y0, args.width, rows)
        if args.interleave:
            stride, offset = (int(v) for v in args.interleave.split(","))
            r.set_row_interleave(stride, offset)
        ctx.append((spec, r, opaque, transparent))
    res = {i: [] for i in range(len(ctx))}
    for rnd in range(args.rounds + 1):
        for i, (path, r, opaque, transparent) in enumerate(ctx):
            r.sync()
            t = time.perf_counter()
            for _ in range(args.frames):
                r.clear_color((1, 1, 1, 1))
                r.draw_geometry(scene, opaque, transparent)
            r.sync()
            if rnd:
                res[i].append((time.perf_counter() - t) / args.frames * 1e3)
    for i, (path, *_rest) in enumerate(ctx):
        a = np.array(res[i])
        print(f"{path}: median {np.median(a):.4f} min {a.min():.4f} max {a.max():.4f} ms/frame")
    if args.stages:  # per-stage kernel times (events on every stage: the pipeline is perturbed, compare like with like)
        for path, r, opaque, transparent in ctx:
            r.set_option(A.OPT_KERNEL_TIMING, 2)
            acc = []
            for _ in range(5):
                for _ in range(20):
                    r.clear_color((1, 1, 1, 1))
                    r.draw_geometry(scene, opaque, transparent)
                r.sync()
                st = r.get_stats()
                acc.append((st.geometry_ms, st.binning_ms, st.tile_ms))
            r.set_option(A.OPT_KERNEL_TIMING, 0)
            print(f"{path}: geometry/binning/tile median {np.median(np.array(acc), axis=0).round(4).tolist()} ms")

    for _path, r, *_rest in ctx:
        r.close()


if __name__ == "__main__":
    main()
