#!/bin/bash
# round 3: cat-free decoder -- op / model tests, then in-step A/B (fill_cat_buffers on / off), fp32 and bf16
out=gpurun_out/r3
mkdir -p $out
python3 -m pytest tests/test_hip_ops.py tests/test_hip_model.py -x -q -m gpu -k "into_cat or filled_cat or module_128 or bf16" 2>&1 | tail -8
for prec in fp32 bf16; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec > $out/cat_fill1_$prec.json 2> $out/e.err
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fill-cat 0 --precision $prec > $out/cat_fill0_$prec.json 2>> $out/e.err
done
python3 - <<PY
import json
for f in ('cat_fill1_fp32','cat_fill0_fp32','cat_fill1_bf16','cat_fill0_bf16'):
    d=json.loads(open('$out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_ms'])
PY
