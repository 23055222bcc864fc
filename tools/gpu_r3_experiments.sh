#!/bin/bash
# Round-3 measurements that are not part of the suite / bench / profile scripts.  One recipe per call:
#     gpurun -- bash tools/gpu_r3_experiments.sh <recipe> [args]
# Recipes (outputs under gpurun_out/r3/):
#   gc            the driver's bench command with and without FlowTrainer's gc.freeze() (host_gc object of the JSON line)
#   tolerances    tools/tolerance_probe.py: how far gradients / bf16 flows are from their references (what the test bars come from)
#   corr_bwd      tools/microbench.py corr_bwd_rs: row-streamed cost-volume backward vs the group-split ring kernel and the
#                 tuning-only variants (UNFLOW_RS_VARIANTS=9,21,22,... : prefetch depths, ablations 21-29, two-half form 12-14,
#                 wave sets 15-17, 16x32 tiles 10); corr8_bwd_rs the same at d = 8
#   instep        bench.py on the tuning library: row-streamed (default) vs group-split (UNFLOW_CORR_BWD=4) inside the train step
#   cat           bench.py with / without the epilogue-filled cat buffers (--fill-cat), fp32 and bf16
#   warp_gather   the gather-form warp backward (unflow_warp_bwd_det) vs the scatter forms, per level and flow kind
#   glue SWITCH   same-box A/B of one of the glue-launch fusions of DESIGN section 3 item 32 / 33: the parity tests that cover it, then the
#                 driver-style bench line with the switch on and off, hipGraph replay and eager.  SWITCH = fused-upsample | fused-loss-sums |
#                 fused-head | dup-centre | weight-shadows (the last one on --precision bf16)
#   aten          tools/probes/aten_sources.py: the ATen / runtime kernels of an eager step by launching op and autograd node (fp32, bf16)
#   entry E L     rocprofv3 kernel trace of one C entry point at one level (tools/pmc_entry.py): per-kernel averages
out=gpurun_out/r3
mkdir -p $out
r=${1:-help}; shift
case $r in
  gc)
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --gc-freeze 0 > $out/bench_gc0.json 2> $out/gc.err
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_gc1.json 2>> $out/gc.err
    python3 -c "
import json
for f in ('bench_gc0','bench_gc1'):
    d=json.loads(open('$out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['value'], d['ms_per_step'], d['step_ms'], d['host_gc'])" ;;
  tolerances) python3 tools/tolerance_probe.py 2>&1 | tee $out/tolerance_probe.txt | tail -40 ;;
  corr_bwd) echo 'round 5: the row-streamed variants this recipe swept were removed from csrc/corr.hip (profiles/r3_experiments.md has their numbers); see tools/gpu_r5.sh mfma_sweep' ;;
  instep)
    python3 tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline > $out/instep_rs.json 2> $out/instep.err
    UNFLOW_CORR_BWD=4 python3 tools/bench_with_lib.py --steps 20 --warmup 5 --no-cpu-baseline > $out/instep_gs.json 2>> $out/instep.err
    python3 -c "
import json
for f in ('instep_rs','instep_gs'):
    d=json.loads(open('$out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], 'aggregate', d['roofline']['aggregate']['us_per_step'], d['roofline']['aggregate']['frac'])
    for e in d['roofline']['aggregate']['per_level']:
        if e['entry']=='unflow_corr_bwd': print('   ', e['shape'], e['avg_us'], e['frac'])" ;;
  cat)
    for prec in fp32 bf16; do for f in 1 0; do
      python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision $prec --fill-cat $f > $out/cat_fill${f}_$prec.json 2>> $out/cat.err
      python3 -c "
import json; d=json.loads(open('$out/cat_fill${f}_$prec.json').read().strip().splitlines()[-1]); print('$prec fill-cat $f', d['value'], d['ms_per_step'], d['step_ms']['median'])"
    done; done ;;
  warp_gather) UNFLOW_MICROBENCH_TUNING=1 timeout 300 python3 tools/microbench.py warp_gather 2>&1 | tee $out/warp_gather.txt | grep warp_bwd ;;
  glue)
    sw=${1:-fused-head}; extra=""; [ $sw = weight-shadows ] && extra="--precision bf16"
    python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "upsample or flow_head or loss_bookkeeping or cat_channels_last" 2>&1 | tail -3
    python3 -m pytest tests/test_hip_model.py -q -m gpu -k "golden or filled or shadows" 2>&1 | tail -3
    for g in -1 0; do for v in 1 0; do
      python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $extra --graph $g --$sw $v > $out/glue_${sw}${v}_graph$g.json 2> $out/glue_${sw}${v}_graph$g.err
    done; done
    python3 -c "
import json,glob
for f in sorted(glob.glob('$out/glue_${sw}*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['value'], d['ms_per_step'], d['step_ms']['median'], d['step_mode'])" ;;
  aten)
    python3 tools/probes/aten_sources.py > $out/aten_sources_fp32.txt 2>&1; python3 tools/probes/aten_sources.py bf16 > $out/aten_sources_bf16.txt 2>&1
    grep total $out/aten_sources_fp32.txt $out/aten_sources_bf16.txt ;;
  entry)
    e=${1:-unflow_warp_bwd_det}; l=${2:-L2}; d=$GRAFT_REPO_ROOT/$out/prof_entry_${e}_$l; mkdir -p $d
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/tools/pmc_entry.py $e $l 10 > $d/run.log 2>&1 )
    rm -f $d/*/*kernel_trace.csv
    python3 -c "
import csv,glob
for r in csv.DictReader(open(glob.glob('$d/*/*kernel_stats.csv')[0])):
    n=r['Name']
    if ('namespace' in n or 'unflow' in n) and 'at::native' not in n: print('   %-64s calls %s avg %.1f us' % (n.replace('(anonymous namespace)::','')[:64], r['Calls'], float(r['AverageNs'])/1e3))" ;;
  *) sed -n 2,16p $0 ;;
esac
