#!/bin/bash
# Developer tool (GPU box): SQ instruction/occupancy counters of one build over a few frames of configs[3].
#   tools/pmc_run.sh TAG [path/to/libsvr_hip.so]      -> gpurun_out/pmc_TAG.txt
# Counters are collected in separate rocprofv3 passes (a pass holds a handful), kernel trace only.
tag=$1; lib=${2:-}
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out/pmc_$tag
libarg=""; [ -n "$lib" ] && libarg="--lib $root/$lib"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES" \
           "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_IFETCH SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  # under `timeout -k`: a set the SQ cannot collect aborts rocprofv3, which then does not exit; a failed set is named in the summary
  if ! timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set -d $out/pmc_$tag/s$i -o s$i -- python3 $root/tools/frames.py --frames 6 $libarg > $out/pmc_$tag/s$i.log 2>&1; then
    echo "set $i failed: $set" | tee -a $out/pmc_$tag/failed.txt
  fi
done
{ python3 $root/tools/pmc_summary.py $(find $out/pmc_$tag -name "*_results.db" | sort); [ -f $out/pmc_$tag/failed.txt ] && sed 's/^/# FAILED /' $out/pmc_$tag/failed.txt; } > $out/pmc_$tag.txt 2>&1
grep -E "^#|tile_kernel<0, false, false>|setup_kernel" $out/pmc_$tag.txt
