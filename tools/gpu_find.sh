# Regenerate / extend the shipped MIOpen find-db: exhaustive find over the conv shapes of the given bench
# configurations.  The db under unopticalflow_amd/miopen_db is copied to gpurun_out (merged back to the
# build container), extended in place by MIOpen, and is then committed by hand.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db
cp unopticalflow_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
export UNFLOW_MIOPEN_FORCE_FIND=1   # (not needed for a user-provided db path; kept for clarity)
date
timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
timeout 1500 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --precision bf16 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
timeout 2400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --hw 448 1024 --batch 4 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
ls -la gpurun_out/miopen_db
