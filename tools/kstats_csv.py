"""Developer tool: per-kernel duration statistics of a rocprofv3 --kernel-trace database as CSV (the layout of
rocprofv3 --stats' kernel_stats.csv).   python tools/kstats_csv.py x_results.db "comment: the command traced" """
import collections
import sqlite3
import statistics
import sys

c = sqlite3.connect(sys.argv[1])
d = collections.defaultdict(list)
for n, s, e in c.execute("select name, start, end from kernels order by start"):
    d[n].append(e - s)
total = sum(sum(v) for v in d.values())
print(f"# {sys.argv[2]} (from the rocpd database, tools/kstats_csv.py)")
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    sd = statistics.pstdev(v) if len(v) > 1 else 0.0
    print(f'"{k}",{len(v)},{sum(v)},{sum(v) / len(v):.1f},{100.0 * sum(v) / total:.2f},{min(v)},{max(v)},{sd:.1f}')
