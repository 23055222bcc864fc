cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp or corr" 2>&1 | tail -5 > gpurun_out/r2/ops_tests_b.log
cat gpurun_out/r2/ops_tests_b.log
UNFLOW_MICROBENCH_TUNING=1 timeout 600 python tools/microbench.py ablate 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/ablate_b.txt
cat gpurun_out/r2/ablate_b.txt
timeout 300 python tools/microbench.py corr 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/corr_b.txt
cat gpurun_out/r2/corr_b.txt
bash tools/gpu_pmc.sh r2/pmc_b pmc_l2 2>&1 | tail -30 > gpurun_out/r2/pmc_b.txt
cat gpurun_out/r2/pmc_b.txt
