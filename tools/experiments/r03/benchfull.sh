#!/bin/bash
# the driver's default bench command, timed
mkdir -p gpurun_out/r03d
S=$(date +%s)
python bench.py > gpurun_out/r03d/bench_default.json 2> gpurun_out/r03d/bench_default.err
echo "rc=$? wall=$(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r03d/bench_default.json"))
print(d["value"], d["ms_per_step"]); print(json.dumps(d.get("other_configs"), indent=1)); print(d["cpu_baseline"]["value"], d["end_to_end"].get("reads_per_s"))
PY
