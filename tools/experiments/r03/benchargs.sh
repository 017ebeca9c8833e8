# GPU box: the bench under argument combinations a driver might pass.  usage: bash tools/experiments/r03/benchargs.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for args in "--gpus 1 --steps 5 --warmup 2" "--gpus 1 --steps 1 --warmup 0" "--gpus 1 --steps 20 --warmup 5" "--steps 3 --warmup 1 --mode se" ; do
  i=$((i+1)); S=$(date +%s)
  timeout 900 python3 bench.py $args > $O/b$i.json 2> $O/b$i.err; rc=$?
  python3 -c "
import json
try:
    d=json.load(open('$O/b$i.json')); print('[$args] rc=$rc wall=%d s: %.2f M reads/s %.1f ms/step steps %d warmup %d keys %s' % ($(date +%s)-$S, d['value']/1e6, d['ms_per_step'], d['steps'], d['warmup'], sorted(k for k in d if k not in ('metric','unit','config','roofline'))))
except Exception as e: print('[$args] rc=$rc FAILED', e); print(open('$O/b$i.err').read()[-1500:])"
done
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
