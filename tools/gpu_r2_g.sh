#!/bin/bash
# cost-volume kernels after a change: parity, then timing (shipped library)
mkdir -p gpurun_out/r2
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "corr" 2>&1 | tail -3
python tools/microbench.py corr corr8 2>&1 | grep -v amdgpu | tee gpurun_out/r2/corr_g.txt
