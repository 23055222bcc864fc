# round 2: tile-shape / variant sweeps on the tuning library + parity of the changed kernels
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp or corr" 2>&1 | tail -15 > gpurun_out/r2/ops_tests.log
cat gpurun_out/r2/ops_tests.log
export UNFLOW_MICROBENCH_TUNING=1
timeout 600 python tools/microbench.py warp_c 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/warp_c_sweep.txt
cat gpurun_out/r2/warp_c_sweep.txt
timeout 600 python tools/microbench.py corr_bwd_sweep 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/corr_bwd_sweep.txt
cat gpurun_out/r2/corr_bwd_sweep.txt
