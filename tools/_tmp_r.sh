out=gpurun_out/r3/final4; mkdir -p $out
python3 -m pytest tests -q -m gpu 2>&1 | tail -2 | tee $out/pytest_gpu.txt
python3 __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
python3 tools/pmc_traffic.py 2>&1 | tail -7
cp gpurun_out/r3/r3_pmc_traffic.json profiles/r3_pmc_traffic.json
for k in a b; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_$k.json 2> $out/e_$k; done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --graph 0 > $out/bench_n1_graph0.json 2> $out/e_c
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > $out/bench_bf16.json 2> $out/e_d
python3 -c "
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    d=json.loads([l for l in open(f).read().splitlines() if l.startswith('{')][-1])
    r=d['roofline']
    print(f.split('/')[-1], d['value'], d['ms_per_step'], d['step_mode'], d['step_ms']['median'], d['step_ms']['max'], r['avg_us'], r['frac'], r['traffic'], r['aggregate']['us_per_step'], r['aggregate']['frac'], d['conv_stack']['frac_lower_bound'])
"
bash tools/gpu_r3_profile.sh fp32 2>&1 | grep -v '^"' | cut -c1-200 | tail -6
