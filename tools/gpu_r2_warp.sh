# round 2: LDS-tile warp kernels -- parity tests + per-level timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp" 2>&1 | tail -15 > gpurun_out/r2/warp_tests.log
cat gpurun_out/r2/warp_tests.log
timeout 300 python tools/microbench.py warp 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/warp_microbench.txt
cat gpurun_out/r2/warp_microbench.txt
