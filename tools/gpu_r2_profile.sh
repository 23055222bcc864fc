# round 2: rocprofv3 kernel stats of the timed region + PMC traffic of the cost-volume / warp entry points + the bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
bash tools/gpu_profile_only.sh r2/prof 2>&1 | tail -14
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/pmc_traffic.py 2>&1 | grep -v amdgpu.ids | tail -8
cp gpurun_out/r2/r2_pmc_traffic.json profiles/r2_pmc_traffic.json 2>/dev/null
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r2/bench_final.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench_final.json').read())
r=d['roofline']; print(d['value'], d['ms_per_step'], '| roofline', r['kernel'][:50], r['avg_us'], r['frac'], 'traffic', r['traffic'], '| agg', r['aggregate']['us_per_step'], r['aggregate']['frac'])
PY
