cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp or corr" 2>&1 | tail -5 > gpurun_out/r2/ops_tests_c.log
cat gpurun_out/r2/ops_tests_c.log
UNFLOW_MICROBENCH_TUNING=1 timeout 600 python tools/microbench.py ablate warp_c 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/ablate_c.txt
cat gpurun_out/r2/ablate_c.txt
