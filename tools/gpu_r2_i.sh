#!/bin/bash
# channels_last conv stacks: epilogue parity, model parity (immediate-mode MIOpen), quick bench
mkdir -p gpurun_out/r2
python -m pytest tests/test_hip_ops.py -x -q -m gpu -k "bias_leaky or conv_block" 2>&1 | tail -5
python -m pytest tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -15
python bench.py --steps 20 --warmup 5 2>&1 | grep "^{" | tee gpurun_out/r2/bench_cl.json | cut -c1-400
