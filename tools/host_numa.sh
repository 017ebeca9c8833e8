# GPU box: does pinning the command line to one NUMA node help its host side?  usage: bash tools/host_numa.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -3
for ts in "" "0-63,128-191" "64-127,192-255" "0-15" "0-7,128-135"; do
  BSX_TASKSET=$ts python3 tools/e2e_bench.py --pairs 16777216 --genome 0.002 --dir /dev/shm/bsx_nu_$$ > $O/${TAG}_numa.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/${TAG}_numa.json')); t=d['timing']; n=2*d['pairs']
print('taskset [$ts]: %.1f M reads/s  mapping %.2f s' % (n/t['mapping_s']/1e6, t['mapping_s']), t['stage_busy_s']['format'], t['stage_busy_s']['write'], t['mapping_cpu_s'])"
done
