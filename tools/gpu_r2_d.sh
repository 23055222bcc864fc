cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp_corr" 2>&1 | tail -8 > gpurun_out/r2/fused_tests.log
cat gpurun_out/r2/fused_tests.log
timeout 300 python tools/microbench.py fused 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/fused_bench.txt
cat gpurun_out/r2/fused_bench.txt
