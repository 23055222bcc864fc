# same-box A/B: default | fused | default | graph | force-ddp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2
for args in "--steps 50" "--fused 1" "--steps 50 --warmup 10" "--graph 1" "--force-ddp" "--fused 1 --graph 1"; do
  timeout 900 python bench.py --no-cpu-baseline $args 2>&1 | grep "^{" | tail -1 > gpurun_out/r2/ab.json
  python3 - "$args" <<'PY'
import json, sys
d = json.loads(open('gpurun_out/r2/ab.json').read())
r = d.get('roofline') or {}
print('%-28s %7.2f pairs/s %7.3f ms/step | agg %s us %s' % (sys.argv[1], d['value'], d['ms_per_step'], (r.get('aggregate') or {}).get('us_per_step'), (r.get('aggregate') or {}).get('frac')))
PY
done
