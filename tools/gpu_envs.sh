# GPU box: serial-mode bench line under several environment settings.  usage: bash tools/gpu_envs.sh <tag> "NAME=V ..." ["NAME=V ..." ...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python3 bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_$i.json 2> $O/${TAG}_$i.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_$i.json")); k = d["roofline"]["dominant_kernel"]
    print("[$e] serial ms/step %.1f  hscan ms/step %.1f  Gcand/s %.1f  launches %.0f" % (d["ms_per_step"], k["ms_per_step"], k["candidates_per_s"] / 1e9, k["launches_per_step"]))
except Exception as e:
    print("[$e] no line:", e)
PY
done
