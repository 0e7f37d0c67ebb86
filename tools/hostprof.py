"""Developer tool: host-side cost of one frame's calls with an idle GPU (every frame is fenced).
    python tools/hostprof.py [--lib build_ab/libsvr_hip_hostprof.so]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default="")
ap.add_argument("--frames", type=int, default=400)
args = ap.parse_args()
pkg = g.load_package()
S, A = pkg.scenes, pkg.abi
hip = A.SvrLib(os.path.abspath(args.lib)) if args.lib else pkg.load_product_library()
sc = S.sponza_like(lod=1, tex_size=64)
W, H = 3840, 2160
r = hip.create(W, H)
handles = sc.upload(r)
opaque, transparent = sc.render_objects(handles)
scene = S.scene_data_struct(*S.config3_camera(), W, H)
r.set_scissor(0, 0, W, 270)
t_clear, t_draw = [], []
for _ in range(args.frames):
    t0 = time.perf_counter()
    r.clear_color((1, 1, 1, 1))
    t1 = time.perf_counter()
    st = r.draw_geometry(scene, opaque, transparent)
    t2 = time.perf_counter()
    r.sync()
    t_clear.append(t1 - t0)
    t_draw.append(t2 - t1)
print(f"idle-GPU host time per call (us): clear_color {np.median(t_clear) * 1e6:.1f}, draw_geometry {np.median(t_draw) * 1e6:.1f} "
      f"(of which inside the library {st.mesh_draw_time * 1e3:.1f})")
