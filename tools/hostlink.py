"""Developer tool: what crosses the host link, and how fast — the one-time scene upload (svr_upload_mesh,
svr_create_image incl. mip generation), the per-frame inputs (already inside every timed frame) and the read-backs.

    python tools/hostlink.py [--width 3840 --height 2160]
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
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    args = ap.parse_args()
    pkg = g.load_package()
    S, A = pkg.scenes, pkg.abi
    hip = pkg.load_product_library()
    sc = S.sponza_like(lod=1, tex_size=1024)
    r = hip.create(args.width, args.height)
    r.sync()
    t0 = time.perf_counter()
    handles = sc.upload(r)
    r.sync()
    t_up = time.perf_counter() - t0
    c = sc.counts()
    geo = c["vertices"] * 48 + c["triangles"] * 12
    tex = sum(int(np.asarray(t).nbytes) for t in sc.textures)
    print(f"scene upload: {geo / 1e6:.1f} MB geometry + {tex / 1e6:.1f} MB texels (level 0; mips are made on the device) in {t_up * 1e3:.1f} ms "
          f"= {(geo + tex) / t_up / 1e9:.2f} GB/s through svr_upload_mesh / svr_create_image")
    opaque, transparent = sc.render_objects(handles)
    pos, pitch, yaw = S.config3_camera()
    scene = S.scene_data_struct(pos, pitch, yaw, args.width, args.height)
    n_obj = len(opaque) + len(transparent)
    print(f"per frame, inside every timed frame: {n_obj} RenderObjects x {C_sizeof(A.SvrRenderObject)} B + 240 B of scene constants = "
          f"{(n_obj * C_sizeof(A.SvrRenderObject) + 240) / 1e3:.1f} KB of host memory read by svr_draw_geometry; "
          f"{n_obj} draw records + wave chunks pulled from pinned staging by the pass's own prologue kernel")
    for _ in range(3):
        r.clear_color((1, 1, 1, 1))
        r.draw_geometry(scene, opaque, transparent)
    r.sync()
    for name, fn, nbytes in (("svr_read_color (RGBA16F)", lambda: r.read_color(), args.width * args.height * 8),
                             ("svr_read_color (as RGBA8)", lambda: r.read_color(as_rgba8=True), args.width * args.height * 4),
                             ("svr_read_depth", lambda: r.read_depth(), args.width * args.height * 4),
                             ("svr_read_swapchain (B8G8R8A8)", lambda: r.read_swapchain(args.width, args.height, 0), args.width * args.height * 4)):
        fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        print(f"{name}: {nbytes / 1e6:.1f} MB in {t * 1e3:.2f} ms = {nbytes / t / 1e9:.2f} GB/s to pageable host memory")
    # a frame whose image is read back every frame (a host consumer): frame + present + read-back
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            r.clear_color((1, 1, 1, 1))
            r.draw_geometry(scene, opaque, transparent)
            r.read_swapchain(args.width, args.height, 0)
        ts.append((time.perf_counter() - t0) / 10)
    t = float(np.median(ts))
    print(f"frame + svr_read_swapchain every frame: {t * 1e3:.2f} ms/frame = {1.0 / t:.0f} frames/s (host-link inclusive)")
    r.close()


def C_sizeof(t):
    import ctypes
    return ctypes.sizeof(t)


if __name__ == "__main__":
    main()
