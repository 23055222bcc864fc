#!/bin/bash
# round 3: per-kernel times of the gather-form warp backward (rocprofv3 kernel trace over the microbenchmark)
out=$GRAFT_REPO_ROOT/gpurun_out/r3/prof_warp_gather
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
UNFLOW_MICROBENCH_TUNING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/microbench.py warp_gather > $out/run.log 2>&1
cd $GRAFT_REPO_ROOT
S=$(ls $out/*/*kernel_stats.csv | head -1)
cp $S $out/kernel_stats.csv
rm -f $out/*/*kernel_trace.csv
grep -E "warp|zero" $out/kernel_stats.csv | cut -c1-220 | head -20
