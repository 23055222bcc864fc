#!/bin/bash
# in-step A/B of the packed FMAs in the backward row pipeline (tuning builds; forward with scalar FMAs in both)
mkdir -p gpurun_out/r2
report() {
python - $1 <<'PY'
import json,sys
d=json.load(open('gpurun_out/r2/ab_%s.json'%sys.argv[1]))
a=d['roofline']['aggregate']
sel=[e for e in a['per_level'] if e['entry'] in ('unflow_corr_fwd','unflow_corr_bwd') and e['shape'][2] in (64,32,16)]
print(sys.argv[1], d['value'], d['ms_per_step'], 'agg', a['us_per_step'], ' | '.join('%s %s %.1f'%(e['entry'][7:],e['shape'][1],e['avg_us']) for e in sorted(sel, key=lambda e:(e['entry'],e['shape'][1]))))
PY
}
for i in 1 2; do python tools/bench_with_lib.py --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r2/ab_bwdpk.json; report bwdpk; done
UNFLOW_TUNING_EXTRA_FLAGS="-DUNFLOW_NO_PK_FWD -DUNFLOW_NO_PK_BWD" python -m unopticalflow_amd.build --tuning --force 2>&1 | grep -E "error|Error"
for i in 1 2; do python tools/bench_with_lib.py --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r2/ab_bwdnopk.json; report bwdnopk; done
