#!/bin/bash
# round 3: bf16 option glue -- the cat / NCHW hand-off with a bf16 NHWC side (unflow_cat_nhwc_bf16 / unflow_split_nhwc_bf16) and one
# multi-tensor weight cast per pass (net_utils.WeightShadows): parity tests, then the bf16 step with and without the shadows
out=gpurun_out/r3/bf16glue
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "cat_channels_last or upsample" 2>&1 | tail -40 | tee $out/pytest_ops.txt
python3 -m pytest tests/test_hip_model.py -q -m gpu -x 2>&1 | tail -3 | tee $out/pytest_model.txt
for g in -1 0; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 --graph $g > $out/bench_bf16_graph$g.json 2> $out/bench_bf16_graph$g.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 --graph $g --weight-shadows 0 > $out/bench_bf16_graph${g}_noshadow.json 2> $out/bench_bf16_graph${g}_noshadow.err
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
