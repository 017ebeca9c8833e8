# GPU box: unit groups on C4 / C5 with the default batches in flight.  usage: bash tools/experiments/r03/g4.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for m in rrbs trim; do
for g in 1 2; do
  BSX_HEAVY_GROUPS=$g python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/${m}_g$g.json 2> $O/${m}_g$g.err
  python3 -c "
import json; d=json.load(open('$O/${m}_g$g.json')); print('$m groups $g: %.1f ms/step %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))"
done
done
