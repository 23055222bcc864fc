# BASELINE configs [2] (bf16 conv stacks, per-GPU batch 8) and [3] (Sintel 1024x448, batch 4) on one GPU, plus the RCCL rehearsal of [1]
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for args in ${UNFLOW_CONFIGS:-"--precision bf16" "--hw 448 1024 --batch 4" "--force-ddp" "--fused 1" "--graph 1"}; do
  tag=$(echo $args | tr -d ' -')
  ( time timeout 900 python bench.py --steps 30 --warmup 10 --no-cpu-baseline $args 2>&1 | grep -v amdgpu.ids | grep "^{" | tail -1 > gpurun_out/r2/bench_$tag.json ) 2>&1 | grep real
  python3 - "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open('gpurun_out/r2/bench_%s.json' % sys.argv[1]).read())
    r = d.get('roofline') or {}
    print(sys.argv[1], d['value'], 'pairs/s', d['ms_per_step'], 'ms/step', '| roofline', (r.get('kernel') or '')[:44], r.get('avg_us'), r.get('frac'), '| agg', (r.get('aggregate') or {}).get('us_per_step'), (r.get('aggregate') or {}).get('frac'))
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/r2/bench_%s.json' % sys.argv[1]).read()[-300:])
PY
done
