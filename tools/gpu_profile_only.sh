cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1b
cd /tmp && export TMPDIR=/tmp
UNFLOW_MIOPEN_FIND=${UNFLOW_MIOPEN_FIND:-1} timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1b/run.log 2>&1
grep -E '^\{' $GRAFT_REPO_ROOT/gpurun_out/prof_r1b/run.log | cut -c1-200
ls -la $GRAFT_REPO_ROOT/unopticalflow_amd/miopen_db/
