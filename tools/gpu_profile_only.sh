cd $GRAFT_REPO_ROOT
OUT=${1:-prof_r1c}
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
UNFLOW_MIOPEN_FIND=${UNFLOW_MIOPEN_FIND:-1} timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $GRAFT_REPO_ROOT/gpurun_out/$OUT/run.log 2>&1
grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/$OUT/run.log | cut -c1-200
cd $GRAFT_REPO_ROOT
T=$(ls gpurun_out/$OUT/*/*kernel_trace.csv | head -1)
python3 tools/summarize_trace.py $T gpurun_out/$OUT/timed_region_stats.csv --steps 9
rm -f $T   # the raw trace is large; the summary is what gets committed
