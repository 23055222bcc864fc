#!/bin/bash
# bench line + per-level table (kernel-exact timing)
mkdir -p gpurun_out/r2
python bench.py ${1:---no-cpu-baseline} 2>&1 | grep "^{" > gpurun_out/r2/bench_exact.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r2/bench_exact.json"))
r=d["roofline"]; a=r["aggregate"]
print(d["value"], d["ms_per_step"], "roofline", r["avg_us"], r["frac"], r["traffic"], "agg", a["us_per_step"], a["frac"])
for e in a["per_level"]: print("  %-16s %-20s %7.2f us %6.0f GB/s %.3f"%(e["entry"],e["shape"],e["avg_us"],e["algorithmic_GBps"],e["frac"]))
for e in d["kernel_survey"]: print("  S %-28s %-20s %7.2f us x%.0f"%(e["entry"],e["shape"],e["avg_us"],e["launches_per_step"]))
PY
