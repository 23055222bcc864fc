#!/bin/bash
# round 3: pyramid hand-off that writes the centre features twice (unflow_to_nchw_dup / unflow_to_nhwc_fold): parity, then the step
out=gpurun_out/r3/dup
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "cat_channels_last or flow_head or loss_bookkeeping" 2>&1 | tail -30 | tee $out/pytest_ops.txt
python3 -m pytest tests/test_hip_model.py -q -m gpu -x -k "golden or filled or shadows or default" 2>&1 | tail -30 | tee $out/pytest_model.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_graph-1.json 2> $out/bench_graph-1.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 > $out/bench_graph0.json 2> $out/bench_graph0.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('step_mode'))
    except Exception as e:
        print(f, 'ERR', e)
PY
