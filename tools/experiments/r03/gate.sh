# GPU box: device batches per GPU and how many of them may compute at once (slots granted in batch order).  usage: bash tools/experiments/r03/gate.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2; do
for cfg in "3 3" "4 2" "4 3" "5 3" "6 3"; do
  set -- $cfg
  BSX_GPU_BATCHES=$1 BSX_GPU_COMPUTE=$2 BSX_TIMING=2 python3 tools/e2e_bench.py --pairs 16777216 --genome 1.0 --dir /dev/shm/bsx_t_$$ > $O/e2e_b$1_c$2_$rep.json 2> $O/e2e_b$1_c$2_$rep.err
  python3 -c "
import json
try:
    d=json.load(open('$O/e2e_b$1_c$2_$rep.json')); t=d['timing']; n=2*d['pairs']
    print('batches $1 computing $2 #$rep e2e: %.2f M reads/s  mapping %.2f s' % (n/t['mapping_s']/1e6, t['mapping_s']), {k: t['stage_busy_s'][k] for k in ('gpu','gpu_align','format','write')})
except Exception as e: print('batches $1 computing $2 failed', open('$O/e2e_b$1_c$2_$rep.err').read()[-400:])"
done
done
python3 tools/e2e_gantt.py $O/e2e_b5_c3_1.json | head -19
