#!/bin/bash
# One GPU-box call of round 1f: full GPU suite, headline bench, rocprofv3 of the two weight-stream
# kernels, bench lines of the other shapes.  Run from the repo root.
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/full_gpu.log 2>&1
tail -3 gpurun_out/full_gpu.log
timeout -k 10 200 python bench.py > gpurun_out/r1f_bench.json || exit 1
bash profiles/run_profile.sh r1f_quadw --weights tag127 > gpurun_out/prof_w.log 2>&1 || exit 1
bash profiles/run_profile.sh r1f_quadl --weights lstmA > gpurun_out/prof_l.log 2>&1 || exit 1
for t in tag133 synA4 tag127 lstmA; do
  timeout -k 10 100 python bench.py --weights $t --no-cpu-baseline > gpurun_out/b_$t.json || exit 1
done
python - <<'PY'
import json
for t in ("tag133", "synA4", "tag127", "lstmA"):
    d = json.load(open("gpurun_out/b_" + t + ".json"))
    print(t, round(d["ms_per_step"], 4), round(d["roofline"]["kernel_ms"], 4), d["roofline"]["kernel"], round(d["roofline"]["achieved"], 2))
d = json.load(open("gpurun_out/r1f_bench.json"))
print("headline", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["cpu_baseline"]["value"])
PY
