cd $GRAFT_REPO_ROOT
for v in 5 1; do
UNFLOW_CORR_VARIANT=$v timeout 200 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k "corr" 2>&1 | tail -3
UNFLOW_CORR_VARIANT=$v timeout 120 python tools/microbench.py corr 2>&1 | grep -v amdgpu | head -3
done
