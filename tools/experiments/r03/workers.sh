# GPU box: format workers of the command line now that the driver threads sleep (tiny genome and full size).  usage: bash tools/experiments/r03/workers.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for g in 0.002 1.0; do
for p in 12 14 16 20; do
  python3 tools/e2e_bench.py --pairs 16777216 --genome $g --threads $p --dir /dev/shm/bsx_w_$$ > $O/w_${g}_$p.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/w_${g}_$p.json')); t=d['timing']; b=t['stage_busy_s']; n=2*d['pairs']
print('genome x$g -p $p: %.1f M reads/s  mapping %.2f s  busy' % (n/t['mapping_s']/1e6, t['mapping_s']), {k: b[k] for k in ('parse','format','write')})"
done
done
