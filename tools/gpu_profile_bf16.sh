cd $GRAFT_REPO_ROOT
OUT=prof_bf16
mkdir -p gpurun_out/$OUT
cd /tmp && export TMPDIR=/tmp
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --precision bf16 > $GRAFT_REPO_ROOT/gpurun_out/$OUT/run.log 2>&1
grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/$OUT/run.log | cut -c1-200
cd $GRAFT_REPO_ROOT
T=$(ls gpurun_out/$OUT/*/*kernel_trace.csv | head -1)
python3 tools/summarize_trace.py $T gpurun_out/$OUT/timed_region_stats.csv --steps 9
rm -f $T
