#!/bin/bash
# round 3, the profile artefacts of one source state (copy what is to be judged from gpurun_out/r3/ into profiles/):
#   prof_<prec>/timed_region_stats.csv   rocprofv3 --kernel-trace over bench.py, last 9 steps       (tools/summarize_trace.py)
#   r3_conv_mfma_<prec>.json             MFMA counters of the conv kernels, separate --pmc pass       (tools/summarize_mfma.py)
#   r3_pmc_traffic.json                  HBM bytes per launch of the cost-volume / warp entry points  (tools/pmc_traffic.py)
#   prof_corr8/kernel_stats.csv          BASELINE configs[4]: d = 8 cost volume on all five levels
# usage: bash tools/gpu_r3_profile.sh [fp32] [bf16] [traffic] [corr8]
cd $GRAFT_REPO_ROOT
what=${@:-fp32 bf16 traffic corr8}
for w in $what; do
  case $w in
    fp32|bf16)
      extra="--graph 0"; [ $w = bf16 ] && extra="--precision bf16 --graph 0"      # (eager: a --pmc pass over hipGraph replays left the GPU unresponsive once)
      OUT=$GRAFT_REPO_ROOT/gpurun_out/r3/prof_$w; mkdir -p $OUT
      ( cd /tmp && export TMPDIR=/tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $OUT/run.log 2>&1 )
      T=$(ls $OUT/*/*kernel_trace.csv | head -1)
      python3 tools/summarize_trace.py $T $OUT/timed_region_stats.csv --steps 9 | head -3
      rm -f $T
      M=$GRAFT_REPO_ROOT/gpurun_out/r3/mfma_$w; mkdir -p $M
      ( cd /tmp && export TMPDIR=/tmp && timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $M -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing $extra > $M/run.log 2>&1 )
      CC=$(ls $M/*/*counter_collection.csv | head -1)
      peak=157.3; [ $w = bf16 ] && peak=2500
      python3 tools/summarize_mfma.py $CC $OUT/timed_region_stats.csv gpurun_out/r3/r3_conv_mfma_$w.json --steps 9 --peak $peak && rm -f $CC
      ;;
    traffic) python3 tools/pmc_traffic.py 2>&1 | tail -8 ;;
    corr8) bash tools/gpu_corr8_profile.sh 2>&1 | tail -14; mkdir -p gpurun_out/r3/prof_corr8; cp gpurun_out/prof_corr8/kernel_stats.csv gpurun_out/prof_corr8/run.log gpurun_out/r3/prof_corr8/ ;;
  esac
done
