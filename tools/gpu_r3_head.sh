#!/bin/bash
# round 3: flow heads as one kernel each way (unflow_flow_head_*), closed-form up-sampling backward for factors 2 / 4, no zero-filled
# gradients for masks / weights: parity, then the step with / without the fused heads
out=gpurun_out/r3/head
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "flow_head or loss_bookkeeping or upsample or warp_golden or warp_mask or occ_weight or losses_vs" 2>&1 | tail -30 | tee $out/pytest_ops.txt
python3 -m pytest tests/test_hip_model.py -q -m gpu -x 2>&1 | tail -30 | tee $out/pytest_model.txt
for g in -1 0; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph $g > $out/bench_graph$g.json 2> $out/bench_graph$g.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph $g --fused-head 0 > $out/bench_graph${g}_atenhead.json 2> $out/bench_graph${g}_atenhead.err
done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('step_mode'))
    except Exception as e:
        print(f, 'ERR', e)
PY
