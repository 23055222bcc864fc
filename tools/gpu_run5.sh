cd $GRAFT_REPO_ROOT
echo base; timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
echo find; UNFLOW_MIOPEN_FIND=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
echo cl; UNFLOW_CHANNELS_LAST=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*, "unit": "pairs/s".*"ms_per_step": [0-9.]*'
