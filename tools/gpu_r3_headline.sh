#!/bin/bash
# round 3, VERDICT item 1: the driver's exact bench command, repeated, next to the variants that separate host pacing from
# device time (no kernel-timing events, hipGraph replay, longer run).  Usage: gpurun -- bash tools/gpu_r3_headline.sh [tag]
tag=${1:-a}
out=gpurun_out/r3
mkdir -p $out
rocm-smi --showclocks --showpower > $out/smi_before_$tag.txt 2>&1
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_${tag}1.json 2> $out/bench_driver_${tag}1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_driver_${tag}2.json 2>> $out/bench_driver_${tag}1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/bench_notiming_${tag}.json 2>> $out/bench_driver_${tag}1.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --graph 1 > $out/bench_graph_${tag}.json 2>> $out/bench_driver_${tag}1.err
python3 bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline > $out/bench_50_${tag}.json 2>> $out/bench_driver_${tag}1.err
rocm-smi --showclocks --showpower > $out/smi_after_$tag.txt 2>&1
python3 - <<PY
import json,glob
for f in sorted(glob.glob('$out/bench_*_${tag}*.json')+glob.glob('$out/bench_*_${tag}.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d['step_ms'], 'enq', d['host_enqueue_ms']['median'], 'drain', d['drain_ms'])
    except Exception as e:
        print(f, 'ERR', e)
PY
