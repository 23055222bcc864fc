# usage: bash tools/gpu_pmc.sh <outdir> <microbench-args...>   (PMC passes only: no trace domains mixed in)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $pmc | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $pmc --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/microbench.py "$@" > $OUT/$tag.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('$OUT/*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k=(r['Kernel_Name'][:60], r['Grid_Size'])
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    if 'corr' in k[0] or 'ssim' in k[0] or 'warp' in k[0]:
        print(k, {c: round(sum(x)/len(x),1) for c,x in sorted(v.items())})
PY
