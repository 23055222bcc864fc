set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_r1
timeout 300 python bench.py --steps 20 --warmup 5 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/bench_r1.log
cat gpurun_out/bench_r1.log
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r1/run.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/prof_r1/run.log
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name "*.csv" | head; 
