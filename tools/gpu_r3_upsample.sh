#!/bin/bash
# round 3: the fused flow up-sampling (unflow_upsample_scaled_*) -- its parity tests, the model tests, and the step with / without it
out=gpurun_out/r3/upsample
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "upsample" 2>&1 | tail -3 | tee $out/pytest_ops.txt
python3 -m pytest tests/test_hip_model.py -q -m gpu -x 2>&1 | tail -3 | tee $out/pytest_model.txt
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1.json 2> $out/bench_n1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 > $out/bench_graph0.json 2> $out/bench_graph0.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-upsample 0 > $out/bench_n1_torch_up.json 2> $out/bench_n1_torch_up.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-upsample 0 --graph 0 > $out/bench_graph0_torch_up.json 2> $out/bench_graph0_torch_up.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('step_mode'))
    except Exception as e:
        print(f, 'ERR', e)
PY
