# GPU box: transfer leg after the scheduling change; three / four batches in flight on C3.  usage: bash tools/experiments/r03/misc.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 > $O/tr.json 2> $O/tr.err
python3 -c "
import json; d=json.load(open('$O/tr.json')); print('default %.1f ms/step; incl transfers' % d['ms_per_step'], json.dumps(d['value_incl_transfers'])[:300])"
for nf in 3 4; do
  python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 --transfer-steps 0 --steps 12 --warmup 4 --in-flight $nf > $O/f$nf.json 2> $O/f$nf.err
  python3 -c "
import json; d=json.load(open('$O/f$nf.json')); print('in flight $nf: %.1f ms/step %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))"
done
