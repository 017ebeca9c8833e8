# GPU box: the -m gpu suite, then a serial-mode and a default bench line.  usage: bash tools/gpu_check.sh <tag> [pytest args]
TAG=${1:-chk}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q "$@" > $O/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${TAG}_pytest.log
timeout 600 python3 bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_bench_serial.json 2> $O/${TAG}_bench_serial.err; echo "serial rc=$?"
python3 - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_bench_serial.json")); k = d["roofline"]["dominant_kernel"]
    print("serial ms/step %.1f  hscan ms/step %.1f  Gcand/s %.1f  launches %.0f" % (d["ms_per_step"], k["ms_per_step"], k["candidates_per_s"] / 1e9, k["launches_per_step"]))
except Exception as e:
    print("no serial line:", e)
PY
