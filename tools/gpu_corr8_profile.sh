# BASELINE config 5: d=8 cost volume on every pyramid level of 832x256 (2B=16), kernel trace + stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_corr8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_corr8 -- python3 $GRAFT_REPO_ROOT/tools/microbench.py corr8 > $GRAFT_REPO_ROOT/gpurun_out/prof_corr8/run.log 2>&1
cd $GRAFT_REPO_ROOT
grep "corr d" gpurun_out/prof_corr8/run.log
S=$(ls gpurun_out/prof_corr8/*/*kernel_stats.csv | head -1)
cp $S gpurun_out/prof_corr8/kernel_stats.csv
rm -f gpurun_out/prof_corr8/*/*kernel_trace.csv
head -12 gpurun_out/prof_corr8/kernel_stats.csv | cut -c1-200
