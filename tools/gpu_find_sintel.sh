# Extend the shipped MIOpen find-db with the fp32 (channels_last) conv configurations of the 1024x448 bs-4 step (BASELINE configs[3]).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/miopen_db gpurun_out/r2
cp unopticalflow_amd/miopen_db/*.txt gpurun_out/miopen_db/ 2>/dev/null
export MIOPEN_USER_DB_PATH=$GRAFT_REPO_ROOT/gpurun_out/miopen_db
date
timeout 2400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing --hw 448 1024 --batch 4 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
date
ls -la gpurun_out/miopen_db
unset MIOPEN_USER_DB_PATH
cp gpurun_out/miopen_db/*.udb.txt gpurun_out/miopen_db/*.ufdb.txt unopticalflow_amd/miopen_db/
timeout 600 python bench.py --no-cpu-baseline --hw 448 1024 --batch 4 2>&1 | grep "^{" > gpurun_out/r2/bench_hw4481024batch4.json; cut -c1-200 gpurun_out/r2/bench_hw4481024batch4.json
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
