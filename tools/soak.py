"""Developer tool: a long unfenced run with a new camera every frame, two alternating target sets and the
whole operation mix (deferred clears, background effects, passes, identity and scaled blits), checked
against the oracle every N frames.  Looks for what single-frame parity tests cannot see: set reuse,
staging reuse, log retirement and deferred-clear bookkeeping going wrong under a host running ahead.

    python tools/soak.py --frames 3000 --check-every 250
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import svr_testlib as T  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    ap.add_argument("--check-every", type=int, default=250)
    ap.add_argument("--width", type=int, default=480)
    ap.add_argument("--height", type=int, default=270)
    ap.add_argument("--queue-caps", type=int, default=0, help="SVR_OPT_QUEUE_CAPS: start the internal queues this small (forces replays)")
    args = ap.parse_args()
    import torch
    pkg = g.load_package()
    hip, ora = pkg.load_product_library(), T.load_oracle()
    S, A = pkg.scenes, pkg.abi
    W, H = args.width, args.height
    sc = S.sponza_like(lod=8, tex_size=64)
    r = hip.create(W, H)
    ro = ora.create(W, H)
    ora.lib.svr_oracle_set_threads(ro.h, 8)
    if args.queue_caps:
        r.set_option(A.OPT_QUEUE_CAPS, args.queue_caps)
    hh, ho = sc.upload(r), sc.upload(ro)
    op, tr = sc.render_objects(hh)
    opo, tro = sc.render_objects(ho)
    dev = torch.device("cuda", 0)
    r.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    targets = [(torch.zeros((H, W, 4), dtype=torch.float16, device=dev), torch.zeros((H, W), dtype=torch.float32, device=dev))
               for _ in range(2)]
    swap = torch.zeros((H, W, 4), dtype=torch.uint8, device=dev)
    rng = np.random.default_rng(5)
    bad = 0
    for f in range(args.frames):
        cam = ((float(rng.uniform(-5, 65)), float(rng.uniform(0.2, 17)), float(rng.uniform(-14, 14))),
               float(rng.uniform(-1.2, 1.2)), float(rng.uniform(0, 6.283)))
        scene = S.scene_data_struct(*cam, W, H)
        c, d = targets[f & 1]
        r.bind_targets(c.data_ptr(), d.data_ptr())
        kind = f % 5
        if kind == 3:
            r.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
        elif kind == 4:
            r.draw_background(A.BACKGROUND_GRADIENT, (0.2, 0.3, 0.9, 1.0, 0.9, 0.4, 0.1, 1.0) + (0.0,) * 8)
        else:
            r.clear_color((1.0, 1.0, 1.0, 1.0))
        r.draw_geometry(scene, op, tr)
        if f % 7 == 0:
            r.copy_to_swapchain(swap.data_ptr(), W, H, A.SWAPCHAIN_B8G8R8A8)
        if (f + 1) % args.check_every == 0:
            got, gotd = r.read_color(), r.read_depth()
            if kind == 3:
                ro.draw_background(A.BACKGROUND_SKY, A.SKY_DEFAULT)
            elif kind == 4:
                ro.draw_background(A.BACKGROUND_GRADIENT, (0.2, 0.3, 0.9, 1.0, 0.9, 0.4, 0.1, 1.0) + (0.0,) * 8)
            else:
                ro.clear_color((1.0, 1.0, 1.0, 1.0))
            ro.draw_geometry(scene, opo, tro)
            ok = np.array_equal(got, ro.read_color()) and np.array_equal(gotd.view(np.uint32), ro.read_depth().view(np.uint32))
            bad += 0 if ok else 1
            print(f"frame {f + 1}: {'identical' if ok else 'DIFFERENT'}", flush=True)
    r.sync()
    print(f"{args.frames} frames, {bad} mismatching checks, replayed passes {r.get_stats().replayed_passes}")
    r.close()
    ro.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
