cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
export UNFLOW_MICROBENCH_TUNING=1
timeout 600 python tools/microbench.py ablate 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/ablate.txt
cat gpurun_out/r2/ablate.txt
