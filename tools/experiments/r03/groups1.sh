# GPU box: unit groups with ONE batch in flight.  usage: bash tools/experiments/r03/groups1.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for m in pe trim rrbs; do
for g in 1 2 3; do
  BSX_HEAVY_GROUPS=$g python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 --transfer-steps 0 --steps 6 --warmup 2 --in-flight 1 > $O/${m}_g$g.json 2> $O/${m}_g$g.err
  python3 -c "
import json; d=json.load(open('$O/${m}_g$g.json')); print('$m groups $g, one in flight: %.1f ms/step %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))"
done
done
