#!/bin/bash
# Developer tool (GPU box): instruction-cache counters of one build over a few frames of configs[3] (tile_kernel is
# 56 KB of code; a pair of CUs shares one 64-KB instruction cache).
#   tools/pmc_icache.sh TAG [path/to/libsvr_hip.so] [frames.py arguments]     -> gpurun_out/pmci_TAG.txt
tag=$1; lib=${2:-}; shift; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out/pmci_$tag
libarg=""; [ -n "$lib" ] && [ "$lib" != "-" ] && libarg="--lib $root/$lib"
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQC_ICACHE_INPUT_VALID_READY SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ"; do
  i=$((i+1))
  if ! timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $out/pmci_$tag/s$i -o s$i -- python3 $root/tools/frames.py --frames 6 $libarg "$@" > $out/pmci_$tag/s$i.log 2>&1; then
    echo "set $i failed: $set" | tee -a $out/pmci_$tag/failed.txt
  fi
done
{ python3 $root/tools/pmc_summary.py $(find $out/pmci_$tag -name "*_results.db" | sort); [ -f $out/pmci_$tag/failed.txt ] && sed 's/^/# FAILED /' $out/pmci_$tag/failed.txt; } > $out/pmci_$tag.txt 2>&1
grep -E "^#|tile_kernel|setup_kernel" $out/pmci_$tag.txt
