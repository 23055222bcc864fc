#!/bin/bash
# Round-6 GPU recipes (the current set, grown out of round 5's; rounds 3 / 4 keep their experiment scripts, their end-of-round scripts are folded in here).
# One or more recipes per call:   gpurun -- bash tools/gpu_r6.sh <recipe> [<recipe> ...]        outputs under gpurun_out/r6/
#   suite        the whole -m gpu suite (no -x) + smoke()
#   headline     the driver's exact bench command, twice
#   configs      the other BASELINE configurations on one GPU: bf16 conv stacks, 1024x448 bs 4, one-rank RCCL, eager
#   ranks8       UNFLOW_BENCH_ONE_GPU=1 python3 bench.py --gpus 8 --batch 2 --steps 5 --warmup 5: eight self-launched ranks on the one GPU
#   profile_fp32 / profile_bf16   rocprofv3 kernel trace of bench.py reduced to the timed steps + the MFMA counters of the conv kernels
#   traffic      tools/pmc_traffic.py (HBM bytes per launch of the cost-volume / warp entry points, separate --pmc passes)
#   corr8        BASELINE configs[4]: rocprofv3 --kernel-trace --stats of microbench corr8 (d = 8 on all five pyramid levels)
#   capi         tools/capi_bench under rocprofv3 --kernel-trace --stats (Python-free capture of the cost-volume kernels)
#   mfma_harness tools/proto/corr_bwd_mfma.hip built and run: matrix-core cost-volume backward variants vs the shipped entry
#   mfma_pmc     SQ counter passes over that harness (three --pmc passes, no trace domains)
#   mfma_sweep   microbench corr_bwd_mf on the tuning library: rows-per-wave sweep at levels 2-4, d = 4 and 8
#   instep_ab    bench.py --corr-bwd {auto, mfma, fp32} and --deferred-loss-sums {0, 1}: in-step A/B of the opt-in switches
#   multiscale   one launch per loss / image warp over the scales (ABI 11): its bit-identity tests, then bench --multiscale-losses 0 / 1
#   smallrows    microbench corr_small: the round-6 small-map cost-volume backward (UNFLOW_CORR_BWD_FP32_NEXT) against today's kernels at levels 5 / 6, d = 4 and 8,
#                its GPU tests, and bench --corr-bwd fp32_next in the step
#   fused_levels bench --fused-levels 4 / 3,4: the fused warp + cost-volume kernel at chosen decoder levels only
#   reopen       what to run FIRST when the lease comes back, most valuable first, so that a cut-off call still leaves the important half:
#                suite, headline, multiscale, instep_ab, smallrows, ranks8, profile_fp32, traffic, fused_levels, configs, profile_bf16, corr8, capi, mfma_harness
#   final        everything that gets recorded for one source state: suite, headline, configs, ranks8, profile_fp32, profile_bf16, traffic, corr8, capi
cd $GRAFT_REPO_ROOT
out=gpurun_out/r6
mkdir -p $out
line() { python3 - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'NO JSON', e); continue
    r = d.get('roofline') or {}
    print(f.split('/')[-1], d['value'], 'pairs/s', d['ms_per_step'], 'ms', '| median', d['step_ms']['median'], 'max', d['step_ms']['max'],
          '| enqueue', d['host_enqueue_ms']['median'], '|', d['step_mode'][:40],
          '| roof', r.get('avg_us'), r.get('frac'), 'agg', (r.get('aggregate') or {}).get('us_per_step'), (r.get('aggregate') or {}).get('frac'),
          'losses', (r.get('losses') or {}).get('us_per_step'), (r.get('losses') or {}).get('frac'), (r.get('losses') or {}).get('launches_per_step'))
PY
}
profile() {   # $1 = fp32 | bf16
  w=$1; extra="--graph 0"; [ $w = bf16 ] && extra="--precision bf16 --graph 0"      # (eager: a --pmc pass over hipGraph replays left the GPU unresponsive once)
  OUT=$GRAFT_REPO_ROOT/$out/prof_$w; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $OUT/run.log 2>&1 )
  T=$(ls $OUT/*/*kernel_trace.csv | head -1)
  python3 tools/summarize_trace.py $T $OUT/timed_region_stats.csv --steps 9 | head -3
  rm -f $T
  M=$GRAFT_REPO_ROOT/$out/mfma_$w; mkdir -p $M
  ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $M -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $M/run.log 2>&1 )
  CC=$(ls $M/*/*counter_collection.csv | head -1)
  peak=157.3; [ $w = bf16 ] && peak=2500
  python3 tools/summarize_mfma.py $CC $OUT/timed_region_stats.csv $out/r6_conv_mfma_$w.json --steps 9 --peak $peak && rm -f $CC
}
for r in "$@"; do
case $r in
  suite)
    timeout 1500 python3 -m pytest tests -m gpu -q -rxX -p no:cacheprovider --timeout 600 --durations=30 > $out/suite.log 2>&1; echo "suite rc=$?"; tail -45 $out/suite.log
    python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
  headline)
    for i in a b; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_$i.json 2> $out/bench_n1_$i.err; done
    line $out/bench_n1_a.json $out/bench_n1_b.json ;;
  configs)
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > $out/bench_precisionbf16.json 2> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hw 448 1024 --batch 4 > $out/bench_hw4481024batch4.json 2>> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-ddp > $out/bench_forceddp.json 2>> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 > $out/bench_n1_graph0.json 2>> $out/configs.err
    line $out/bench_precisionbf16.json $out/bench_hw4481024batch4.json $out/bench_forceddp.json $out/bench_n1_graph0.json ;;
  ranks8)
    UNFLOW_BENCH_ONE_GPU=1 timeout 1500 python3 bench.py --gpus 8 --batch 2 --steps 5 --warmup 5 > $out/bench_8ranks_onegpu.json 2> $out/bench_8ranks_onegpu.err
    echo "ranks8 rc=$?"; line $out/bench_8ranks_onegpu.json; tail -3 $out/bench_8ranks_onegpu.err
    # ... and the test of the same path incl. a killed rank, in a pytest process of its own (skipped inside the whole suite)
    UNFLOW_RUN_EIGHT_RANKS=1 timeout 1800 python3 -m pytest tests/test_cli.py -q -m gpu -p no:cacheprovider -k eight_ranks > $out/ranks8_test.log 2>&1; echo "ranks8 test rc=$?"; tail -5 $out/ranks8_test.log ;;
  profile_fp32) profile fp32 ;;
  profile_bf16) profile bf16 ;;
  traffic) python3 tools/pmc_traffic.py 2>&1 | tail -8 ;;
  corr8) bash tools/gpu_corr8_profile.sh 2>&1 | tail -14; mkdir -p $out/prof_corr8; cp gpurun_out/prof_corr8/kernel_stats.csv gpurun_out/prof_corr8/run.log $out/prof_corr8/ ;;
  capi)
    /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/capi_bench.cpp -Iinclude -Lunopticalflow_amd -lunflow_hip -Wl,-rpath,$GRAFT_REPO_ROOT/unopticalflow_amd -o $out/capi_bench || exit 1
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_capi -- $GRAFT_REPO_ROOT/$out/capi_bench 16 32 64 208 4 50 > $GRAFT_REPO_ROOT/$out/capi_bench.txt 2>&1 )
    tail -8 $out/capi_bench.txt; S=$(ls $out/prof_capi/*/*kernel_stats.csv | head -1); cp $S $out/capi_corr_kernel_stats.csv; head -5 $S; rm -f $out/prof_capi/*/*kernel_trace.csv ;;
  mfma_harness)
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/proto/corr_bwd_mfma.hip -o $out/corr_bwd_mfma || exit 1
    timeout 200 $out/corr_bwd_mfma 16 32 64 208 30 2>&1 | tee $out/corr_bwd_mfma.txt | grep -v "bad" ;;
  mfma_pmc)
    /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Iinclude tools/proto/corr_bwd_mfma.hip -o $out/corr_bwd_mfma || exit 1
    OUT=$GRAFT_REPO_ROOT/$out/mfma_pmc; mkdir -p $OUT
    for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
               "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
               "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
      tag=$(echo $pmc | cut -d' ' -f1)
      ( cd /tmp && export TMPDIR=/tmp && timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/$tag -- $GRAFT_REPO_ROOT/$out/corr_bwd_mfma 16 32 64 208 3 > $OUT/$tag.log 2>&1 )
    done
    python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'][:90], r['Grid_Size'], r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', ''))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as o:
    for k, v in sorted(agg.items()):
        o.write('%s | %s\n' % (k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())}))
print(open('$OUT/summary.txt').read()[:3000])
PY
    ;;
  mfma_sweep) UNFLOW_MICROBENCH_TUNING=1 timeout 600 python3 tools/microbench.py corr_bwd_mf 2>&1 | tee $out/corr_bwd_mf_tuning.txt | tail -60 ;;
  instep_ab)
    for m in auto mfma fp32; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --corr-bwd $m > $out/ab_corr_bwd_$m.json 2>> $out/ab.err; done
    for v in 1 0; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --deferred-loss-sums $v > $out/ab_deferred_loss_sums_$v.json 2>> $out/ab.err; done
    line $out/ab_*.json ;;
  multiscale)
    # one launch per loss over the three scales (csrc/multiscale.h, written after the lease closed): its bit-identity tests, then the step A/B
    timeout 900 python3 -m pytest tests/test_zz_round5_gpu.py -q -m gpu -p no:cacheprovider -k "multiscale or handoff" > $out/multiscale_tests.log 2>&1; echo "multiscale tests rc=$?"; tail -4 $out/multiscale_tests.log
    for v in 0 1; do python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --multiscale-losses $v > $out/ab_multiscale_losses_$v.json 2>> $out/ab.err; done
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --split-handoff 1 > $out/ab_split_handoff_1.json 2>> $out/ab.err
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --split-handoff 1 --multiscale-losses 1 > $out/ab_split_handoff_1_multiscale_1.json 2>> $out/ab.err
    line $out/ab_multiscale_losses_0.json $out/ab_multiscale_losses_1.json $out/ab_split_handoff_1.json $out/ab_split_handoff_1_multiscale_1.json
    python3 - <<'PY'
import json
for v in (0, 1):
    try:
        d = json.loads(open('gpurun_out/r6/ab_multiscale_losses_%d.json' % v).read().strip().splitlines()[-1])
        l = d['roofline']['losses']
        print('multiscale_losses=%d: %.1f pairs/s, loss section %.1f us in %d launches, frac %.3f' % (v, d['value'], l['us_per_step'], l['launches_per_step'], l['frac']))
    except Exception as e:
        print('multiscale_losses=%d: no line (%s)' % (v, e))
PY
    ;;
  smallrows)
    timeout 300 python3 tools/microbench.py corr_small 2>&1 | tee $out/corr_small_microbench.txt | tail -20
    timeout 600 python3 -m pytest tests/test_zz_round5_gpu.py -q -m gpu --runxfail -p no:cacheprovider -k "rows_through_registers" > $out/smallrows_tests.log 2>&1; echo "fp32_next tests rc=$?"; tail -4 $out/smallrows_tests.log
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --corr-bwd fp32_next > $out/ab_corr_bwd_fp32_next.json 2>> $out/ab.err
    line $out/ab_corr_bwd_fp32_next.json ;;
  fused_levels)
    # the fused warp + cost-volume kernel at chosen levels only (it lost with all of 2-4 fused; level 5's width 26 is not served)
    for lv in none 4 3,4; do
      a=""; [ $lv != none ] && a="--fused-levels $lv"
      python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline $a > $out/ab_fused_levels_$lv.json 2>> $out/ab.err
    done
    line $out/ab_fused_levels_*.json ;;
  reopen) bash tools/gpu_r6.sh suite headline multiscale instep_ab smallrows ranks8 profile_fp32 traffic fused_levels configs profile_bf16 corr8 capi mfma_harness ;;
  final) bash tools/gpu_r6.sh suite headline configs ranks8 profile_fp32 profile_bf16 traffic corr8 capi ;;
  *) echo "unknown recipe $r" ;;
esac
done
