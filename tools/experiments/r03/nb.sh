# GPU box: device batches per GPU in the command line (BSX_GPU_BATCHES).  usage: bash tools/experiments/r03/nb.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for nb in 2 3 4; do
  BSX_GPU_BATCHES=$nb python3 tools/e2e_bench.py --pairs 16777216 --genome 1.0 --dir /dev/shm/bsx_nb_$$ > $O/nb${nb}_$rep.json 2> $O/nb${nb}_$rep.err
  python3 -c "
import json
d=json.load(open('$O/nb${nb}_$rep.json')); t=d['timing']; n=2*d['pairs']
print('device batches $nb #$rep: %.2f M reads/s  mapping %.2f s' % (n/t['mapping_s']/1e6, t['mapping_s']), t['stage_busy_s'])"
done
done
