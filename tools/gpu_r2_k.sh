#!/bin/bash
# in-step A/B of the two warp-backward forms (tuning library: UNFLOW_WARP_BWD=0 LDS accumulator, 1 cell gather)
mkdir -p gpurun_out/r2
report() {
python - $1 <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2/ab_%s.json'%sys.argv[1]))
a=d['roofline']['aggregate']
sel=[e for e in a['per_level'] if e['entry'] in ('unflow_warp_bwd',) and e['shape'][1] > 3]
print(sys.argv[1], d['value'], d['ms_per_step'], 'agg', a['us_per_step'], ' | '.join('%s %s %.1f'%(e['entry'][7:],e['shape'][1],e['avg_us']) for e in sorted(sel, key=lambda e:(e['entry'],e['shape'][1]))))
PY
}
for i in 1 2; do
  UNFLOW_WARP_BWD=0 python tools/bench_with_lib.py --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r2/ab_tile.json; report tile
  UNFLOW_WARP_BWD=1 python tools/bench_with_lib.py --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r2/ab_cell.json; report cell
done
