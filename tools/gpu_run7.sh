cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python bench.py --steps 30 --warmup 5 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/bench_r1.log; cat gpurun_out/bench_r1.log | cut -c1-300
timeout 200 python tools/microbench.py corr warp losses 2>&1 | grep -v amdgpu > gpurun_out/microbench_r1.log; cat gpurun_out/microbench_r1.log
