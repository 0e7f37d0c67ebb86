"""Developer tool: per-kernel means of rocprofv3 --pmc results, from counter_collection.csv files or the
rocpd sqlite databases rocprofv3 writes by default (values summed over a dispatch's dimensions).
    python tools/pmc_summary.py gpurun_out/r01g_fetch/f_results.db [more.db|.csv ...]"""
import collections
import csv
import sqlite3
import sys


def rows(fn):
    if fn.endswith(".db"):
        c = sqlite3.connect(fn)
        per, name = collections.defaultdict(float), {}
        for d, k, cn, v in c.execute("select dispatch_id, kernel_name, counter_name, value from counters_collection"):
            per[(d, cn)] += v
            name[d] = k
        for (d, cn), v in per.items():
            yield name[d], cn, v
    else:
        for r in csv.DictReader(open(fn)):
            yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])


for fn in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for k, cn, v in rows(fn):
        agg[k.split("(")[0][:48]][cn].append(v)
    print(f"# {fn}")
    for k, d in sorted(agg.items()):
        cols = ", ".join(f"{c}={sum(v) / len(v):.1f}" for c, v in sorted(d.items()))
        print(f"{k:50s} launches={len(next(iter(d.values()))):4d}  {cols}")
