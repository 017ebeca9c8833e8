# GPU box: C4 with three batches in flight and smaller pools.  usage: bash tools/experiments/r03/rrbs3.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for cfg in "2 0" "3 110000,1400000" "3 85000,1048576" "2 110000,1400000"; do
  set -- $cfg
  HL=""; [ "$2" != "0" ] && HL="--heavy-limits $2"
  timeout 600 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --in-flight $1 $HL > $O/f$1_$2.json 2> $O/f$1_$2.err
  python3 -c "
import json
try:
    d=json.load(open('$O/f$1_$2.json')); print('rrbs in flight $1 limits $2: %.1f ms/step %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))
except Exception as e: print('rrbs in flight $1 limits $2 failed:', open('$O/f$1_$2.err').read()[-300:])"
done
