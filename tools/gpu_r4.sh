#!/bin/bash
# Round-4 GPU recipes.  One or more recipes per call:   gpurun -- bash tools/gpu_r4.sh <recipe> [<recipe> ...]
# Outputs under gpurun_out/r4/ (copy what is to be judged into profiles/r4_*).
#   newtests   the tests this round added or touched (-x, 10-minute cap per test)
#   suite      the whole -m gpu suite (no -x: every failure is listed) + smoke()
#   headline   the driver's exact bench command, twice
#   ranks      `python3 bench.py --gpus 2` with NO launcher in front, two ranks sharing the GPU over gloo (UNFLOW_BENCH_ONE_GPU=1)
#   stepmode   multi-rank step mode A/B on the one-rank RCCL rehearsal (--force-ddp): hipGraph replay + one all-reduce + Adam graph
#              against eager + hooks, fp32 and bf16, on an idle host and confined to ONE core next to a busy-loop neighbour
#   finddb     tools/probes/finddb_run_to_run.py: what two forms of the network differ by under MIOpen's measured picks
#   losses     tools/probes/loss_kernel_times.py + microbench losses
#   capi       build tools/capi_bench and run it under rocprofv3 --kernel-trace --stats (Python-free cost-volume capture)
#   profile    rocprofv3 kernel trace of bench.py reduced to the timed steps (+ MFMA counters): fp32
#   traffic    tools/pmc_traffic.py (HBM bytes per launch, separate --pmc passes)
#   configs    bf16 / 1024x448 bs 4 / one-rank RCCL / eager bench lines
#   rerank     bench A/B of find-db variants in which the runner-up solver is ranked first where it is within 3 / 10 / 30 us (tools/miopen_rerank.py)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r4
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
          '| enqueue', d['host_enqueue_ms']['median'], '| drain', d['drain_ms'], '|', d['step_mode'][:40],
          '| roof', r.get('avg_us'), r.get('frac'), 'agg', (r.get('aggregate') or {}).get('us_per_step'), (r.get('aggregate') or {}).get('frac'),
          'losses', (r.get('losses') or {}).get('us_per_step'), (r.get('losses') or {}).get('frac'))
PY
}
for r in "$@"; do
case $r in
  newtests)   # what this round added or touched, first failure stops, no test may take more than 10 minutes
    timeout 1500 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 600 tests/test_hip_ops.py -k "ssim or losses or reductions or deferred or bias_leaky or flow_head" > $out/newtests_ops.log 2>&1; echo "ops rc=$?"; tail -25 $out/newtests_ops.log
    timeout 1500 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 600 tests/test_abi.py tests/test_cli.py tests/test_hip_model.py -k "c_program or bench_starts or graph_capture_keeps or replayed or flow_adam or kitti_256 or batch8 or sintel or hipgraph or rccl or module_128" > $out/newtests_model.log 2>&1; echo "model rc=$?"; tail -40 $out/newtests_model.log ;;
  smooth)
    timeout 600 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 300 tests/test_hip_ops.py -k "smooth or losses or reductions or golden" > $out/smooth_tests.log 2>&1; echo "smooth tests rc=$?"; tail -8 $out/smooth_tests.log
    timeout 900 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 800 tests/test_hip_model.py -k "flow_adam or find_db or module_128" > $out/smooth_model.log 2>&1; echo "model rc=$?"; tail -8 $out/smooth_model.log
    timeout 300 python3 tools/probes/loss_kernel_times.py 2>&1 | tee $out/loss_kernel_times_b.txt | grep -E "smooth|ssim" ;;
  warpfused)
    timeout 900 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 600 tests/test_hip_ops.py -k "warp" > $out/warpfused_tests.log 2>&1; echo "warp tests rc=$?"; tail -12 $out/warpfused_tests.log
    for v in 1 0; do python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused-warp-bwd $v > $out/warpfused_$v.json 2> $out/warpfused_$v.err; done
    line $out/warpfused_1.json $out/warpfused_0.json
    python3 - <<'PY'
import json
for v in (1, 0):
    d = json.loads(open('gpurun_out/r4/warpfused_%d.json' % v).read().strip().splitlines()[-1])
    for e in d['roofline']['aggregate']['per_level']:
        if 'warp_bwd' in e['entry'] and e['shape'][1] > 3: print(v, e['entry'], e['shape'], e['avg_us'], e['frac'])
PY
    ;;
  resttests)
    timeout 900 python3 -m pytest -x -q -m gpu -p no:cacheprovider --timeout 600 tests/test_hip_model.py -k "graph_capture_keeps or flow_adam or hipgraph or rccl or two_ranks" > $out/resttests.log 2>&1; echo "resttests rc=$?"; tail -30 $out/resttests.log ;;
  suite)
    timeout 2400 python3 -m pytest tests -m gpu -q -p no:cacheprovider --timeout 900 > $out/suite.log 2>&1; echo "suite rc=$?"; tail -15 $out/suite.log
    python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
  headline)
    for i in a b; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_n1_$i.json 2> $out/bench_n1_$i.err; done
    line $out/bench_n1_a.json $out/bench_n1_b.json ;;
  ranks)
    UNFLOW_BENCH_ONE_GPU=1 python3 bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $out/bench_2ranks_onegpu.json 2> $out/bench_2ranks_onegpu.err
    echo "ranks rc=$?"; line $out/bench_2ranks_onegpu.json; tail -3 $out/bench_2ranks_onegpu.err ;;
  stepmode)
    for prec in fp32 bf16; do for g in 1 0; do for c in "" "--contended-host"; do
      tag=idle; [ -n "$c" ] && tag=contended
      timeout 400 python3 bench.py --force-ddp --graph $g --precision $prec --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing $c > $out/stepmode_${prec}_graph${g}_$tag.json 2>> $out/stepmode.err
    done; done; done
    line $out/stepmode_*.json ;;
  finddb) timeout 900 python3 tools/probes/finddb_run_to_run.py > $out/finddb_run_to_run.json 2> $out/finddb.err; cat $out/finddb_run_to_run.json | python3 -m json.tool ;;
  losses) timeout 300 python3 tools/probes/loss_kernel_times.py 2>&1 | tee $out/loss_kernel_times.txt | tail -30 ;;
  capi)
    /opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/capi_bench.cpp -Iinclude -Lunopticalflow_amd -lunflow_hip -Wl,-rpath,$GRAFT_REPO_ROOT/unopticalflow_amd -o $out/capi_bench || exit 1
    ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/prof_capi -- $GRAFT_REPO_ROOT/$out/capi_bench 16 32 64 208 4 50 > $GRAFT_REPO_ROOT/$out/capi_bench.txt 2>&1 )
    cat $out/capi_bench.txt | tail -8; S=$(ls $out/prof_capi/*/*kernel_stats.csv | head -1); cp $S $out/capi_corr_kernel_stats.csv; head -5 $S; rm -f $out/prof_capi/*/*kernel_trace.csv ;;
  profile)
    w=fp32; extra="--graph 0"
    OUT=$GRAFT_REPO_ROOT/$out/prof_$w; mkdir -p $OUT
    ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $OUT/run.log 2>&1 )
    T=$(ls $OUT/*/*kernel_trace.csv | head -1)
    python3 tools/summarize_trace.py $T $OUT/timed_region_stats.csv --steps 9 | head -3
    rm -f $T
    M=$GRAFT_REPO_ROOT/$out/mfma_$w; mkdir -p $M
    ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $M -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $M/run.log 2>&1 )
    CC=$(ls $M/*/*counter_collection.csv | head -1)
    python3 tools/summarize_mfma.py $CC $OUT/timed_region_stats.csv $out/r4_conv_mfma_$w.json --steps 9 --peak 157.3 && rm -f $CC ;;
  traffic) python3 tools/pmc_traffic.py 2>&1 | tail -8 ;;
  ssim_variants)
    # variant builds of the library (made on the build host: UNFLOW_TUNING_TAG=<tag> UNFLOW_TUNING_EXTRA_FLAGS="-D..." python -m
    # unopticalflow_amd.build --tuning), each timed by the loss-kernel probe; the shipped library first
    python3 -m pytest tests/test_hip_ops.py -q -m gpu -k "ssim or losses or reductions" -p no:cacheprovider 2>&1 | tail -3
    echo "== shipped"; timeout 200 python3 tools/probes/loss_kernel_times.py 2>&1 | grep -E "ssim" | tee $out/ssim_shipped.txt
    for lib in unopticalflow_amd/libunflow_hip_tuning_*.so; do
      echo "== $lib"; UNFLOW_LIB_PATH=$GRAFT_REPO_ROOT/$lib timeout 200 python3 tools/probes/loss_kernel_times.py 2>&1 | grep -E "ssim" | tee $out/ssim_$(basename $lib .so).txt
    done ;;
  configs)  # the other BASELINE configurations on one GPU: bf16 conv stacks, 1024x448 bs 4, the one-rank RCCL path (all replayed)
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --precision bf16 > $out/bench_precisionbf16.json 2> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --hw 448 1024 --batch 4 > $out/bench_hw4481024batch4.json 2>> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-ddp > $out/bench_forceddp.json 2>> $out/configs.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 > $out/bench_n1_graph0.json 2>> $out/configs.err
    line $out/bench_precisionbf16.json $out/bench_hw4481024batch4.json $out/bench_forceddp.json $out/bench_n1_graph0.json ;;
  rerank)   # MIOpen picks re-ranked: the runner-up solver first wherever it is within X us of the split-K asm kernel (fwd / bwd-data)
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/rerank_base.json 2> $out/rerank.err
    for x in 3 10 30; do
      python3 tools/miopen_rerank.py --write /tmp/unflow_db_$x --margin-us $x --dirs F,B
      MIOPEN_USER_DB_PATH=/tmp/unflow_db_$x python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/rerank_fb_$x.json 2>> $out/rerank.err
    done
    python3 tools/miopen_rerank.py --write /tmp/unflow_db_w --margin-us 3 --dirs W
    MIOPEN_USER_DB_PATH=/tmp/unflow_db_w python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/rerank_w_3.json 2>> $out/rerank.err
    python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing > $out/rerank_base2.json 2>> $out/rerank.err
    line $out/rerank_*.json ;;
  *) echo "unknown recipe $r" ;;
esac
done
