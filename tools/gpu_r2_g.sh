#!/bin/bash
# group-split backward with buffer-descriptor gather + DMA: parity, stamps, timing
mkdir -p gpurun_out/r2
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "corr" 2>&1 | tail -3
UNFLOW_MICROBENCH_TUNING=1 python tools/microbench.py stamps 2>&1 | grep -v amdgpu | tee gpurun_out/r2/stamps_gs3.txt
python tools/microbench.py corr corr8 2>&1 | grep -v amdgpu | tee gpurun_out/r2/corr_g.txt
