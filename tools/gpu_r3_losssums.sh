#!/bin/bash
# round 3: loss bookkeeping as one launch each way (unflow_loss_combine_*, unflow_weighted_mean_sum_*): parity, then the step with / without
out=gpurun_out/r3/losssums
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "loss_bookkeeping or upsample or cat_channels" 2>&1 | tail -30 | tee $out/pytest_ops.txt
python3 -m pytest tests/test_hip_model.py tests/test_data_parallel.py -q -m gpu -x 2>&1 | tail -30 | tee $out/pytest_model.txt
for g in -1 0; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph $g > $out/bench_graph$g.json 2> $out/bench_graph$g.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph $g --fused-loss-sums 0 > $out/bench_graph${g}_eagersums.json 2> $out/bench_graph${g}_eagersums.err
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('step_mode'))
    except Exception as e:
        print(f, 'ERR', e)
PY
