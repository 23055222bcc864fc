#!/bin/bash
# round 3: GC A/B on the driver's command, tolerance probe, model-level tests in both memory formats
out=gpurun_out/r3
mkdir -p $out
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --gc-freeze 0 > $out/bench_gc0_b.json 2> $out/b.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_gc1_b1.json 2>> $out/b.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_gc1_b2.json 2>> $out/b.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_gc*_b*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['step_ms'], 'enq', d['host_enqueue_ms'], 'drain', d['drain_ms'], d['host_gc'])
    except Exception as e:
        print(f, 'ERR', e)
PY
python3 tools/tolerance_probe.py > $out/tolerance_probe.txt 2>&1
tail -40 $out/tolerance_probe.txt
python3 -m pytest tests/test_hip_model.py -x -q -m gpu 2>&1 | tail -15
