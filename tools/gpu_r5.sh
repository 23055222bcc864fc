# Round-5 GPU recipes (one gpurun call each):  bash tools/gpu_r5.sh <recipe> [args]
#   mfma_pmc <outdir> [harness args]   SQ counter passes over tools/proto/corr_bwd_mfma (shipped fp32 kernel + MFMA variants), no trace domains
#   mfma_trace <outdir> [harness args] rocprofv3 --kernel-trace --stats over the same program
cd $GRAFT_REPO_ROOT
recipe=$1; shift
case $recipe in
mfma_pmc)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
  mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16" \
             "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE"; do
    tag=$(echo $pmc | cut -d' ' -f1)
    timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/$tag -- $GRAFT_REPO_ROOT/tools/proto/corr_bwd_mfma "$@" > $OUT/$tag.log 2>&1
  done
  python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:90], r['Grid_Size'], r.get('VGPR_Count',''), r.get('Accum_VGPR_Count',''), r.get('LDS_Block_Size',''))
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt','w') as o:
    for k,v in sorted(agg.items()):
        line = '%s | %s\n' % (k, {c: round(sum(x)/len(x),1) for c,x in sorted(v.items())})
        o.write(line); print(line, end='')
PY
  ;;
mfma_trace)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
  mkdir -p $OUT
  cd /tmp && export TMPDIR=/tmp
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $GRAFT_REPO_ROOT/tools/proto/corr_bwd_mfma "$@" > $OUT/trace.log 2>&1
  find $OUT/trace -name '*kernel_stats.csv' | head -1 | xargs cat | head -20 | tee $OUT/kernel_stats.csv
  ;;
*) echo "unknown recipe $recipe"; exit 2;;
esac
