"""Developer tool: per-kernel start/end times of a few frames from a rocprofv3 --kernel-trace database, as a
text timeline (who overlaps whom).   python tools/timeline.py gpurun_out/tl/x_results.db [first_frame n_frames]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 6
count = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table' or type='view'")]
kd = [t for t in tabs if t.startswith("kernels")] or [t for t in tabs if "kernel_dispatch" in t]
rows = None
for t in ("kernels",) + tuple(kd):
    try:
        rows = db.execute(f"select name, start, end, queue_id from {t} order by start").fetchall()
        break
    except sqlite3.Error:
        continue
if rows is None:
    print("tables:", tabs)
    sys.exit(1)
rows = [(n.split("(")[0].replace("void svr::", "").replace("svr::", "")[:34], s, e, q) for n, s, e, q in rows]
tiles = [i for i, r in enumerate(rows) if r[0].startswith("prologue")]
if len(tiles) <= first + count:
    first, count = max(0, len(tiles) - 3), 2
t0 = rows[tiles[first]][1]
for n, s, e, q in rows[tiles[first]:tiles[first + count]]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:7.1f})  q{q}  {n}")
