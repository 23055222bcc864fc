# Extend the shipped MIOpen find-db with the bf16 channels_last conv configurations of the 832x256 bs-8 step, then compare layouts.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db gpurun_out/r2
cp unopticalflow_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
date
timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --precision bf16 --channels-last 1 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
for cl in 1 0 1 0; do
  timeout 600 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --precision bf16 --channels-last $cl 2>&1 | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['conv_memory_format'])"
done
ls -la gpurun_out/miopen_db
