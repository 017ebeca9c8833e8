# GPU box: output through a shared mapping against pwrite (host-only ceiling and full size), then the byte-identity tests.  usage: bash tools/experiments/r03/mapwrite.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for cfg in "mmap 8" "pwrite 0" "mmap 4" "mmap 14"; do
  set -- $cfg
  BSX_WRITE=$1 BSX_WRITE_THREADS=$2 python3 tools/e2e_bench.py --pairs 16777216 --genome 0.002 --dir /dev/shm/bsx_m_$$ > $O/hc_$1_$2_$rep.json 2> $O/hc_$1_$2_$rep.err
  python3 -c "
import json
d=json.load(open('$O/hc_$1_$2_$rep.json')); t=d['timing']; n=2*d['pairs']
print('$1 threads $2 #$rep: %.1f M reads/s  mapping %.2f s' % (n/t['mapping_s']/1e6, t['mapping_s']), {k: t['stage_busy_s'][k] for k in ('parse','format','write')}, t['mapping_cpu_s'])"
done
done
python3 -m pytest tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -2
