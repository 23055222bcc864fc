# Extend the shipped MIOpen find-db with the fp32 (channels_last) conv configurations of the 832x256 bs-8 step only.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db
cp unopticalflow_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
date
timeout 2400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>&1 | grep "^{" > gpurun_out/r2/bench_cl_found.json; cut -c1-300 gpurun_out/r2/bench_cl_found.json
ls -la gpurun_out/miopen_db
