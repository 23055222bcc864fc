#!/bin/bash
# round 3: gather-form warp backward: oracle tests, then timing against the scatter forms (tuning library)
out=gpurun_out/r3
mkdir -p $out
timeout 600 python3 -m pytest tests/test_hip_ops.py -x -q -m gpu -k "warp or into_cat" 2>&1 | tail -6
UNFLOW_MICROBENCH_TUNING=1 timeout 300 python3 tools/microbench.py warp_gather > $out/warp_gather.txt 2>&1
grep warp_bwd $out/warp_gather.txt
