"""Developer tool: per-kernel durations and the tail of the dispatch timeline from a rocprofv3
--kernel-trace results database (rocpd sqlite).   python tools/kstats.py gpurun_out/prof_c/c_results.db [n_tail]"""
import collections
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
d = collections.defaultdict(list)
for n, s, e in rows:
    d[n.split("(")[0][:60]].append((e - s) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} n={len(v):5d} med={statistics.median(v):8.1f}us mean={statistics.mean(v):8.1f}us sum={sum(v) / 1e3:8.2f}ms")
tail = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if tail:
    last = rows[-tail:]
    t0 = last[0][1]
    for n, s, e in last:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {n.split('(')[0][:50]}")
