#!/bin/bash
# round 3: the whole GPU suite + smoke + the driver's bench line
out=gpurun_out/r3
mkdir -p $out
python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8
python3 __graft_entry__.py smoke 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_full.json 2> $out/full.err
python3 - <<PY
import json
d=json.loads(open('$out/bench_full.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['step_ms'], d['host_gc'])
r=d['roofline']; print(r['kernel'], r['avg_us'], r['frac'], r['traffic'], 'aggregate', r['aggregate']['us_per_step'], r['aggregate']['frac'])
for e in r['aggregate']['per_level']: print('  ', e['entry'], e['shape'], e['avg_us'], e['frac'])
print(d['cpu_baseline'])
PY
