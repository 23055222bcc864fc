#!/bin/bash
# same-box A/B of Model_flow.dup_centre (two rounds each, interleaved)
out=gpurun_out/r3/dup_ab
mkdir -p $out
for r in 1 2; do for d in 1 0; do for g in -1 0; do
  python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --graph $g --dup-centre $d > $out/bench_dup${d}_graph${g}_$r.json 2> $out/bench_dup${d}_graph${g}_$r.err
done; done; done
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d.get('step_mode'))
    except Exception as e:
        print(f, 'ERR', e)
PY
