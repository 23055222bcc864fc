cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db gpurun_out/miopen_cache
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
export MIOPEN_CUSTOM_CACHE_DIR=$GRAFT_REPO_ROOT/gpurun_out/miopen_cache
date
UNFLOW_MIOPEN_FIND=1 timeout 1600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*' > gpurun_out/find_bench.log
date
cat gpurun_out/find_bench.log
du -sh gpurun_out/miopen_db gpurun_out/miopen_cache
ls gpurun_out/miopen_db | head
# second run: should now be fast and show steady-state
UNFLOW_MIOPEN_FIND=1 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
