# GPU box: host side of `bsmap --lanes` against the number of lanes.  A 6 Mb genome makes the GPU stage nearly free, so the reads/s are what
# the host side (parse, format, write) delivers under the box's CPU quota; then the same at full size.  usage: bash tools/lanes_scaling.sh <tag> [pairs]
TAG=${1:-lanes}; PAIRS=${2:-16777216}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for g in 0.002 1.0; do
  if [ $g = 1.0 ]; then V=";-G 0,0 --lanes --lane-files"; else V=";--lanes=1;--lanes=2;--lanes=3;--lanes=4;--lanes=4 --lane-files;--lanes=8"; fi
  python3 tools/e2e_bench.py --pairs $PAIRS --genome $g --dir /dev/shm/bsx_ln_$$ --variants "$V" > $O/${TAG}_$g.json 2> $O/${TAG}_$g.err
  python3 -c "
import json
d=json.load(open('$O/${TAG}_$g.json'))
for r in d.get('variants', [d]):
    t=r['timing']; n=2*r['pairs']; c=t['mapping_cpu_s']
    print('genome x$g [%-28s] mapping %.2f s = %5.1f M reads/s | cpu user+sys %.1f s = %.2f us/read | join %.2f s | workers %s' % (r.get('cli_args',''), t['mapping_s'], n/t['mapping_s']/1e6, c['user']+c['sys'], (c['user']+c['sys'])/n*1e6, t.get('join_s', 0.0), t.get('workers')))"
done
