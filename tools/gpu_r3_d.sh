#!/bin/bash
# round 3: in-step A/B of the cost-volume backward (tuning library): row-streamed (default) vs group-split ring (UNFLOW_CORR_BWD=4)
out=gpurun_out/r3
mkdir -p $out
python3 tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline > $out/instep_rs.json 2> $out/d.err
UNFLOW_CORR_BWD=4 python3 tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline > $out/instep_gs.json 2>> $out/d.err
python3 - <<PY
import json
for f in ('instep_rs','instep_gs'):
    d=json.loads(open('$out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['step_ms']['median'], 'aggregate', d['roofline']['aggregate']['us_per_step'], d['roofline']['aggregate']['frac'])
    for r in d['roofline']['aggregate']['per_level']:
        if r['entry']=='unflow_corr_bwd': print('   ', r['shape'], r['avg_us'], r['frac'])
PY
UNFLOW_MICROBENCH_TUNING=1 timeout 200 python3 tools/microbench.py corr8_bwd_rs > $out/corr8_bwd_rs.txt 2>&1
cat $out/corr8_bwd_rs.txt | grep corr_bwd
