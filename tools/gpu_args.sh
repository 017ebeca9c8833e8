# GPU box: bench lines for several argument sets.  usage: bash tools/gpu_args.sh <tag> "<args>" ["<args>" ...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
i=0
for a in "$@"; do
  i=$((i+1))
  timeout 600 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 $a > $O/${TAG}_$i.json 2> $O/${TAG}_$i.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_$i.json")); k = d["roofline"]["dominant_kernel"]
    print("[$a] ms/step %.1f  %.2f M reads/s  hscan ms/step %.1f  event ms %.1f" % (d["ms_per_step"], d["value"] / 1e6, k["ms_per_step"], d["roofline"]["event_ms_per_do_batch"]))
except Exception as e:
    print("[$a] no line:", e)
PY
done
