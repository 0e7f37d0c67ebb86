#!/bin/bash
# Developer tool (GPU box): texture-addresser / L1 / L2 counters of the tile kernel over a few frames of configs[3].
#   tools/pmc_mem.sh TAG [frames.py args]     -> gpurun_out/pmc_mem_TAG.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/pmcm_$tag
rm -rf $out; mkdir -p $out
i=0
# two counters of a block per pass: more than the block has ("exceeds the capabilities of the hardware") aborts rocprofv3, which then
# does not exit — hence also the timeout around every pass
for set in "TA_TA_BUSY GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES" \
           "TCP_TOTAL_CACHE_ACCESSES TCP_TCC_READ_REQ" "TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES" \
           "TCC_HIT TCC_MISS" "TCC_REQ TCC_EA0_RDREQ"; do
  i=$((i+1))
  echo "pass $i: $set" >> $root/gpurun_out/pmc_mem_$tag.progress
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $out/s$i -o s$i -- python3 $root/tools/frames.py --frames 6 --timing 0 "$@" > $out/s$i.log 2>&1 || echo "set $i failed" >> $root/gpurun_out/pmc_mem_$tag.progress
done
{ python3 $root/tools/pmc_summary.py $(find $out -name "*_results.db" | sort); grep failed $root/gpurun_out/pmc_mem_$tag.progress | sed 's/^/# FAILED /'; } > $root/gpurun_out/pmc_mem_$tag.txt 2>&1
grep -E "^#|tile_kernel<0, false, false>" $root/gpurun_out/pmc_mem_$tag.txt
rm -rf $out
