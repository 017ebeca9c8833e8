# GPU box: serial-mode and default-mode (two batches in flight) bench lines for the default library and for each named variant.  usage: bash tools/gpu_variants.sh <tag> [variant ...]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in default "$@"; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_$v.json 2> $O/${TAG}_$v.err
  python3 - <<PY
import json
try:
    d = json.load(open("$O/${TAG}_$v.json")); k = d["roofline"]["dominant_kernel"]
    print("$v: serial ms/step %.1f  hscan ms/step %.1f  Gcand/s %.1f  launches %.0f  event_ms %.1f" % (d["ms_per_step"], k["ms_per_step"], k["candidates_per_s"] / 1e9, k["launches_per_step"], d["roofline"]["event_ms_per_do_batch"]))
except Exception as e:
    print("$v: no line:", e)
PY
  timeout 600 python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 > $O/${TAG}_${v}_default.json 2>> $O/${TAG}_$v.err
  python3 -c "import json;d=json.load(open('$O/${TAG}_${v}_default.json'));print('$v: default ms/step %.1f  reads/s %.0f' % (d['ms_per_step'], d['value']))"
done
