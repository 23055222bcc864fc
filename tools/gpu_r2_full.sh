# round 2: whole GPU suite + smoke + bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r2/gpu_tests.log
cat gpurun_out/r2/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python bench.py 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r2/bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2/bench.json').read())
print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')
r=d['roofline']; print('roofline:', r['kernel'], r['avg_us'], 'us', r['frac'], 'traffic', r['traffic'], r['traffic_source'])
a=r['aggregate']; print('aggregate:', a['us_per_step'], 'us/step', a['frac'])
for e in a['per_level']: print('  %-22s %-22s %7.2f us  %6.0f GB/s' % (e['entry'], e['shape'], e['avg_us'], e['algorithmic_GBps'] or 0))
print('cpu:', d['cpu_baseline']['value'], 'conv:', d['conv_stack']['frac_lower_bound'])
for e in d['kernel_survey']: print('  S %-26s %-22s %7.2f us x%.0f' % (e['entry'], e['shape'], e['avg_us'], e['launches_per_step']))
PY
