#!/bin/bash
# cpu_baseline leg only: short bench, every other leg off
mkdir -p gpurun_out/r03d
python bench.py --steps 4 --warmup 2 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 > gpurun_out/r03d/bench_cpub.json 2> gpurun_out/r03d/bench_cpub.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03d/bench_cpub.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d["cpu_baseline"], indent=1))
PY
