#!/bin/bash
# GPU box: everything profiles/<tag>_* is made from, for the build in the tree.   tools/profile_round.sh r03_h
#   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace (per-kernel durations of bench.py's own command line)
#   <tag>_pmc.txt            rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_*), one counter group per run, kernel trace only
#   <tag>_traffic.json / <tag>_valu.json   what bench.py quotes in roofline.traffic / roofline_valu
#   <tag>_solo_kernel_stats.csv            the same kernels with the two stages serialised (SVR_OPT_TUNING bit 1): solo durations
#   <tag>_bench.json         the bench line itself (run last, unprofiled)
tag=${1:-r03_h}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out/$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
CMD="python3 $root/bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline"
# Every rocprofv3 pass runs under `timeout -k`: a counter set a block cannot collect ("exceeds the capabilities of
# the hardware") aborts rocprofv3 inside the first HIP call and the aborted process then does not exit.  The program
# itself follows `--` directly (no env / bash -c hop under the profiler's preload).  A pass that fails is NAMED in
# <tag>_pmc.txt instead of being dropped silently.  The SQ sets hold at most 8 counters (the SQ block's limit here);
# FETCH_SIZE and WRITE_SIZE (TCC) go alone, as the guide's HBM section prescribes.
failed=""
pass() {  # pass NAME [--pmc counters...] -- program...
  local name=$1; shift
  if ! timeout -k 10 240 rocprofv3 --kernel-trace "$@" > $out/$name.log 2>&1; then
    failed="$failed $name"
    echo "profile_round: pass $name FAILED (rc $?)" | tee -a $out/failed.txt
  fi
}
pass kt -d $out/kt -o kt -- $CMD
pass f --pmc FETCH_SIZE -d $out/f -o f -- $CMD
pass w --pmc WRITE_SIZE -d $out/w -o w -- $CMD
pass q1 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $out/q1 -o q1 -- $CMD
pass q2 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $out/q2 -o q2 -- $CMD
pass solo -d $out/solo -o solo -- python3 $root/tools/frames.py --frames 30 --timing 0 --tuning 2
cd $root
db() { find $out/$1 -name "*_results.db" | head -1; }
python3 tools/kstats_csv.py $(db kt) "rocprofv3 --kernel-trace -- python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline" > $out/${tag}_kernel_stats.csv
python3 tools/kstats_csv.py $(db solo) "rocprofv3 --kernel-trace -- python tools/frames.py --frames 30 --timing 0 --tuning 2 (geometry + binning and tiles serialised: solo durations)" > $out/${tag}_solo_kernel_stats.csv
{ echo "# rocprofv3 --kernel-trace --pmc <counters> -- python bench.py --steps 20 --warmup 5 --blocks 3 --no-cpu-baseline  (separate passes; FETCH_SIZE/WRITE_SIZE in KiB per launch, SQ_* quad-cycles / instructions per launch; tools/pmc_summary.py)"; python3 tools/pmc_summary.py $(db f) $(db w) $(db q1) $(db q2); [ -n "$failed" ] && echo "# FAILED passes (no counters from them):$failed"; } > $out/${tag}_pmc.txt
python3 tools/profile_json.py $out/${tag}_pmc.txt $out/${tag}_kernel_stats.csv $tag $out
cp $out/${tag}_traffic.json $out/${tag}_valu.json $root/profiles/  # (the box's copy of the tree: bench.py quotes them)
python3 bench.py --profile-tag $tag > $out/${tag}_bench.json 2> $out/bench.err
rm -rf $out/kt $out/f $out/w $out/q1 $out/q2 $out/solo
ls -la $out; tail -c 1500 $out/${tag}_bench.json
