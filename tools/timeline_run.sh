#!/bin/bash
# GPU box: kernel trace of a few frames -> gpurun_out/timeline_TAG.txt      tools/timeline_run.sh TAG [frames.py args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
root=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf $root/gpurun_out/tl_$tag; mkdir -p $root/gpurun_out/tl_$tag
rocprofv3 --kernel-trace -d $root/gpurun_out/tl_$tag -o x -- python3 $root/tools/frames.py --frames 12 "$@" > $root/gpurun_out/tl_$tag/run.log 2>&1
db=$(find $root/gpurun_out/tl_$tag -name "*_results.db" | head -1)
python3 $root/tools/timeline.py $db 8 2 > $root/gpurun_out/timeline_$tag.txt 2>&1
cat $root/gpurun_out/timeline_$tag.txt
rm -rf $root/gpurun_out/tl_$tag
