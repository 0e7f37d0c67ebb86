"""Developer tool: <tag>_traffic.json and <tag>_valu.json (what bench.py quotes) from <tag>_pmc.txt and
<tag>_kernel_stats.csv.   python tools/profile_json.py pmc.txt kernel_stats.csv tag outdir"""
import csv
import json
import re
import sys

pmc, stats, tag, out = sys.argv[1:5]
KERNEL = "tile_kernel<0, false, false>"
vals = {}
for line in open(pmc):
    if KERNEL in line:
        for k, v in re.findall(r"(\w+)=([\d.]+)", line):
            vals[k] = float(v)
avg_ns = None
for row in csv.reader(l for l in open(stats) if not l.startswith("#")):
    if row and KERNEL in row[0]:
        avg_ns = float(row[3])
fetch, write = vals["FETCH_SIZE"], vals["WRITE_SIZE"]
json.dump({"kernel": "tile_kernel<RGBA16F, uninstrumented, whole tiles>", "workload": "bench.py default (configs[3], 3840x2160, 1 GPU)",
           "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
           "correction": "gfx950: FETCH_SIZE counts 128-B read requests at 64 B -> doubled (MI355X_MICROARCH.md, HBM section; calibrated there for "
                         "16-B-per-lane streaming reads, this kernel mixes 4/8/16-B gathers, so the doubled figure is an upper bound); WRITE_SIZE taken as is",
           "traffic_bytes_per_launch": int(2 * fetch * 1024 + write * 1024), "source": f"profiles/{tag}_pmc.txt (two separate rocprofv3 --pmc passes)"},
          open(f"{out}/{tag}_traffic.json", "w"), indent=1)
simd_quad_cycles = 1024 * avg_ns * 2.4 / 4.0  # 256 CUs x 4 SIMDs, 2.4 GHz, a wave64 VALU instruction occupies its SIMD for 4 cycles
json.dump({"kernel": "tile_kernel<RGBA16F, uninstrumented, whole tiles>", "bound": "valu issue", "insts": vals["SQ_INSTS_VALU"],
           "active_quad_cycles": vals["SQ_ACTIVE_INST_VALU"], "avg_launch_ns": avg_ns,
           "issue_frac": vals["SQ_ACTIVE_INST_VALU"] / simd_quad_cycles,
           "wave_cycles": vals.get("SQ_WAVE_CYCLES"), "wait_any": vals.get("SQ_WAIT_ANY"), "wait_inst_any": vals.get("SQ_WAIT_INST_ANY"),
           # resident waves per SIMD, averaged over the kernel: wave-residence quad-cycles against the SIMDs' quad-cycles
           "mean_waves_per_simd": (vals["SQ_WAVE_CYCLES"] / simd_quad_cycles) if vals.get("SQ_WAVE_CYCLES") else None,
           "definition": "SQ_ACTIVE_INST_VALU (quad-cycles the SIMDs spent issuing VALU) / (1024 SIMDs x the kernel's average duration at 2.4 GHz / 4)",
           "source": f"profiles/{tag}_pmc.txt + profiles/{tag}_kernel_stats.csv"},
          open(f"{out}/{tag}_valu.json", "w"), indent=1)
print(open(f"{out}/{tag}_valu.json").read())
