#!/bin/bash
# round 3: everything that gets recorded for one source state -- the GPU suite, smoke, the driver's bench line (twice), the other
# configurations, then the profile artefacts (tools/gpu_r3_profile.sh).  Outputs under gpurun_out/r3/final/.
out=gpurun_out/r3/final
mkdir -p $out
python3 -m pytest tests -q -m gpu 2>&1 | tail -5 | tee $out/pytest_gpu.txt
python3 __graft_entry__.py smoke 2>&1 | tail -1 | tee $out/smoke.txt
for k in a b; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_$k.json 2> $out/bench_n1_$k.err; done
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > $out/bench_bf16.json 2> $out/bench_bf16.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hw 448 1024 --batch 4 > $out/bench_hw4481024batch4.json 2> $out/bench_sintel.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 > $out/bench_graph0.json 2> $out/bench_graph0.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 --graph 0 > $out/bench_bf16_graph0.json 2> $out/bench_bf16_graph0.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-ddp > $out/bench_forceddp.json 2> $out/bench_forceddp.err
UNFLOW_BENCH_ONE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank.err
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d['value'], d['ms_per_step'], d.get('step_ms',{}).get('median'), d['config']['conv_memory_format'], (d['roofline'] or {}).get('frac'))
    except Exception as e:
        print(f, 'ERR', e)
PY
bash tools/gpu_r3_profile.sh ${@:-fp32 bf16 traffic corr8}
