#!/bin/bash
# warp backward, cell-gather form: parity, then the sweep against the LDS-accumulator form (tuning library)
mkdir -p gpurun_out/r2
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp" 2>&1 | tail -3
UNFLOW_MICROBENCH_TUNING=1 python tools/microbench.py warp_c 2>&1 | grep -v amdgpu | tee gpurun_out/r2/warp_h.txt
