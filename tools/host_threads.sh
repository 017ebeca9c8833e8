# GPU box: the command line's host side (tiny genome) under several -p settings.  usage: bash tools/host_threads.sh <tag> "6 8 10 12 16"
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for p in $2; do
  python3 tools/e2e_bench.py --pairs 16777216 --genome 0.002 --threads $p --dir /dev/shm/bsx_ht_$$ > $O/${TAG}_ht_$p.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/${TAG}_ht_$p.json')); t=d['timing']; b=t['stage_busy_s']; n=2*d['pairs']
print('-p $p: %.1f M reads/s  mapping %.2f s  busy' % (n/t['mapping_s']/1e6, t['mapping_s']), {k: b[k] for k in ('parse','format','write')}, 'cpu', t.get('mapping_cpu_s'))"
done
