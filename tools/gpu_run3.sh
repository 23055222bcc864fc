cd $GRAFT_REPO_ROOT
timeout 200 python -m pytest tests/test_hip_ops.py -m gpu -x -q -k corr 2>&1 | tail -3
for v in 1 2 3 4; do UNFLOW_CORR_VARIANT=$v timeout 120 python tools/microbench.py corr 2>&1 | grep -v amdgpu; done
