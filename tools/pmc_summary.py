"""Developer tool: per-kernel means of rocprofv3 --pmc counter_collection.csv files.
    python tools/pmc_summary.py gpurun_out/r01d_fetch/f_counter_collection.csv [more.csv ...]"""
import collections
import csv
import sys

for fn in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"].split("(")[0][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"# {fn}")
    for k, d in sorted(agg.items()):
        cols = ", ".join(f"{c}={sum(v) / len(v):.1f}" for c, v in sorted(d.items()))
        print(f"{k:50s} launches={len(next(iter(d.values()))):4d}  {cols}")
