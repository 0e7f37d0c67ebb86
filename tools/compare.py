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
        a, b = both(T.render_sponza, w, h, lod=args.lod, tex_size=args.tex, instrument=True)
        report("config3", a, b)


if __name__ == "__main__":
    main()
