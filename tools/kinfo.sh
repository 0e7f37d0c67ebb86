#!/bin/bash
# Developer tool: register / LDS / scratch use of every kernel of one source file, from the device assembly.
#   tools/kinfo.sh k_tile [-DFLAG ...]     (keeps /tmp/kinfo_<name>.s for reading)
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/simple-vk-renderer_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math --cuda-device-only -S "$@" "$src/$name.hip" -o /tmp/kinfo_$name.s || exit 1
python3 - /tmp/kinfo_$name.s <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
meta = txt[txt.index("amdhsa.kernels:"):]
for blk in meta.split("- .agpr_count")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    nm = subprocess.run(["c++filt", g("name")], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
    print(f"{nm[:70]:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} vspill {g('vgpr_spill_count'):>3s} sspill {g('sgpr_spill_count'):>3s} "
          f"scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
