# Experiment: MIOpen parameter search (MIOPEN_FIND_ENFORCE=SEARCH) for the tunable solvers on the step's
# conv shapes, time-boxed.  Results accumulate in gpurun_out/miopen_db_tuned/.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db_tuned
cp unopticalflow_amd/miopen_db/*.txt gpurun_out/miopen_db_tuned/
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db_tuned
date
MIOPEN_FIND_ENFORCE=3 timeout ${TUNE_SECONDS:-2400} python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
ls -la gpurun_out/miopen_db_tuned
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
