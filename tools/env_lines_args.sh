# like env_lines.sh with extra bench args.  usage: bash tools/env_lines_args.sh <tag> <mode> <in-flight> "<bench args>" "ENV..." ...
TAG=$1; M=$2; NF=$3; ARGS=$4; shift 4
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python3 bench.py --mode $M --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --steps 6 --warmup 2 --in-flight $NF $ARGS > $O/c_$i.json 2> $O/c_$i.err
  python3 -c "
import json
try:
    d=json.load(open('$O/c_$i.json')); k=d['roofline']['dominant_kernel']; print('$M f$NF $ARGS [$e]: %.1f ms/step  %.2f M reads/s | scan %.1f ms/step %d launches' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], k['launches_per_step']))
except Exception as e: print('$M [$e] failed', e)"
done
