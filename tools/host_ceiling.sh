# GPU box: what the host side of the command line can do when the GPU stage is (nearly) free — a tiny genome — beside the full-size run.
# usage: bash tools/host_ceiling.sh <tag> [pairs]
TAG=${1:-hc}; PAIRS=${2:-16777216}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for g in 0.002 1.0; do
  python3 tools/e2e_bench.py --pairs $PAIRS --genome $g --dir /dev/shm/bsx_hc_$$ > $O/${TAG}_hc_$g.json 2> $O/${TAG}_hc_$g.err
  python3 -c "
import json
d=json.load(open('$O/${TAG}_hc_$g.json')); t=d['timing']; b=t['stage_busy_s']; n=2*d['pairs']
print('genome x$g: mapping %.2f s = %.1f M reads/s | stage busy s:' % (t['mapping_s'], n/t['mapping_s']/1e6), b, '| stage rates M reads/s: parse %.1f format %.1f write %.1f gpu %.1f | workers' % (n/b['parse']/1e6, n/b['format']/1e6, n/b['write']/1e6, n/b['gpu']*2/1e6), t.get('workers'))"
done
