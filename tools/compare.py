"""Developer tool: render a config with the HIP library and the oracle, report where they differ."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as g  # noqa: E402
import svr_testlib as T  # noqa: E402


def report(name, a, b):
    for key in ("color", "depth"):
        x, y = a[key], b[key]
        same = np.array_equal(x.view(np.uint8), y.view(np.uint8))
        if same:
            print(f"  {name}.{key}: identical")
            continue
        d = (x != y)
        if d.ndim == 3:
            d = d.any(axis=2)
        ys, xs = np.nonzero(d)
        print(f"  {name}.{key}: {d.sum()} of {d.size} pixels differ; first {list(zip(xs[:6].tolist(), ys[:6].tolist()))}")
        for px, py in list(zip(xs[:4], ys[:4])):
            print(f"     ({px},{py}) hip={x[py, px]} oracle={y[py, px]}")
    sa, sb = a["stats"], b["stats"]
    print(f"  stats hip: raster={sa.rasterized_fragments} shaded={sa.shaded_fragments} binned={sa.binned_triangles} "
          f"entries={sa.bin_entries} gpu_ms={sa.gpu_time_ms:.3f} | oracle: raster={sb.rasterized_fragments} "
          f"shaded={sb.shaded_fragments} binned={sb.binned_triangles}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="256x144")
    ap.add_argument("--lod", type=int, default=8)
    ap.add_argument("--tex", type=int, default=64)
    ap.add_argument("--configs", default="1,2,3")
    ap.add_argument("--trace", default="", help="x,y: dump the fragment-shader intermediates of that pixel")
    args = ap.parse_args()
    w, h = [int(v) for v in args.size.split("x")]
    pkg = g.load_package()
    hip = pkg.load_product_library()
    ora = T.load_oracle()

    def both(fn, *a, **k):
        out = []
        for lib in (hip, ora):
            t = time.time()
            out.append(fn(lib, *a, **k))
            print(f"  {lib.backend}: {time.time() - t:.3f}s")
        return out

    if "1" in args.configs:
        print("config 1")
        a, b = both(T.render_config1, 256, instrument=True)
        report("config1", a, b)
    if "2" in args.configs:
        print("config 2")
        a, b = both(T.render_config2, w, h, instrument=True)
        report("config2", a, b)
    if "3" in args.configs:
        print("config 3")
        trace = tuple(int(v) for v in args.trace.split(",")) if args.trace else None
        a, b = both(T.render_sponza, w, h, lod=args.lod, tex_size=args.tex, instrument=True, trace=trace)
        report("config3", a, b)
        if trace:
            names = {0: "key", 1: "b1", 2: "b2", 3: "r", 4: "u", 5: "v", 6: "dudx", 7: "dvdx", 8: "dudy", 9: "dvdy",
                     11: "tex.r", 12: "tex.g", 13: "tex.b", 14: "tex.a", 15: "nx", 16: "ny", 17: "nz", 18: "col.r",
                     19: "col.g", 20: "col.b", 21: "light", 22: "out.r", 23: "out.g", 24: "out.b", 25: "out.a",
                     26: "hb1", 27: "hb2", 28: "vb1", 29: "vb2", 30: "hr", 31: "vr", 32: "dst.r", 33: "dst.g",
                     34: "dst.b", 35: "dst.a", 36: "bl.r", 37: "bl.g", 38: "bl.b", 39: "bl.a"}
            ta, tb = a["trace"], b["trace"]
            for i, nm in names.items():
                flag = "" if ta[i].tobytes() == tb[i].tobytes() else "   <-- differs"
                print(f"   {nm:6s} hip={float(ta[i])!r:24} oracle={float(tb[i])!r:24}{flag}")


if __name__ == "__main__":
    main()
