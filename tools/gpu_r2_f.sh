cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1200 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "corr or ssim" 2>&1 | tail -5 > gpurun_out/r2/tests_f.log
cat gpurun_out/r2/tests_f.log
timeout 300 python tools/microbench.py corr 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/corr_f.txt
cat gpurun_out/r2/corr_f.txt
