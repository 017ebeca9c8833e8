# quick default-mode lines under env settings.  usage: bash tools/env_lines.sh <tag> <mode> <in-flight> "ENV1=a ENV2=b" "ENV1=c" ...
TAG=$1; M=$2; NF=$3; shift 3
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for e in "$@"; do
  i=$((i+1))
  env $e timeout 600 python3 bench.py --mode $M --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --steps 6 --warmup 2 --in-flight $NF > $O/b_$i.json 2> $O/b_$i.err
  python3 -c "
import json
try:
    d=json.load(open('$O/b_$i.json')); k=d['roofline']['dominant_kernel']; print('$M f$NF [$e]: %.1f ms/step  %.2f M reads/s | scan %.1f ms/step %d launches' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], k['launches_per_step']))
except Exception as e: print('$M [$e] failed', e)"
done
