cd $GRAFT_REPO_ROOT
echo default-heuristics; date; UNFLOW_MIOPEN_FIND=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
echo shipped-db; date; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
echo graph; date; timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 1 2>&1 | grep -v amdgpu | tail -3 | cut -c1-400
date
