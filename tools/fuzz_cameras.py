"""Developer tool: seeded camera fuzz at a size where heavy tiles are split into quarters (960x540, full geometry), the
HIP frame against the oracle's, bit for bit.   python tools/fuzz_cameras.py [--count 60] [--seed 1]"""
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
    ap.add_argument("--count", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--width", type=int, default=960)
    ap.add_argument("--height", type=int, default=540)
    ap.add_argument("--tuning", type=int, default=None, help="SVR_OPT_TUNING of the HIP side (64: the hierarchical depth test in every pass)")
    ap.add_argument("--instances", type=int, default=1, help="16: BASELINE configs[4]'s 4x4 instancing (deep bins: the hierarchical depth test's filter windows)")
    ap.add_argument("--lod", type=int, default=1)
    ap.add_argument("--plain", action="store_true", help="the HIP frame uninstrumented (the kernels that are timed); fragment counts are then not compared")
    args = ap.parse_args()
    pkg = g.load_package()
    hip, ora = pkg.load_product_library(), pkg.abi.SvrLib(os.path.join(ROOT, "oracle", "libsvr_oracle.so"))
    rng = np.random.default_rng(args.seed)
    inst = pkg.scenes.config5_instances() if args.instances == 16 else None
    bad = 0
    for k in range(args.count):
        pos = (float(rng.uniform(-5, 65)), float(rng.uniform(0.2, 17)), float(rng.uniform(-14, 14)))
        if inst is not None:  # above and around the grid of instances, looking in
            pos = (float(rng.uniform(-20, 160)), float(rng.uniform(2, 40)), float(rng.uniform(-30, 30)))
        if k % 3 == 0 and inst is None:  # inside a curtain's plane: clipped transparent triangles in deep tiles
            pos = (float(rng.uniform(2, 50)), float(rng.uniform(8.5, 15)), float(rng.choice([-6.0, 6.0]) + rng.uniform(-0.2, 0.2)))
        cam = (pos, float(rng.uniform(-1.2, 1.2)), float(rng.uniform(0, 6.283)))
        a = T.render_sponza(hip, args.width, args.height, lod=args.lod, tex_size=64, camera=cam, instances=inst, instrument=not args.plain, tuning=args.tuning)
        b = T.render_sponza(ora, args.width, args.height, lod=args.lod, tex_size=64, camera=cam, instances=inst, instrument=True, threads=16)
        same = all(np.array_equal(a[x], b[x]) for x in ("color", "depth", "rgba8")) and a["stats"].culled_draws == b["stats"].culled_draws and \
            (args.plain or a["stats"].rasterized_fragments == b["stats"].rasterized_fragments)
        bad += not same
        print(f"camera {k}: {'identical' if same else 'DIFFERENT'} {cam}", flush=True)
    print(f"{args.count} cameras, {bad} mismatching")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
