# GPU box: A/B of two builds of libbsx.so on the same box, alternating.  usage: bash tools/ab_libs.sh <tag> <other-lib-name> [modes] [reps]
TAG=$1; OTHER=$2; MODES=${3:-"pe rrbs"}; REPS=${4:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for m in $MODES; do
  for rep in $(seq 1 $REPS); do
    for v in default $OTHER; do
      if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
      timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 12 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
      python3 -c "
import json
try:
    d=json.load(open('$O/${m}_${v}_$rep.json')); k=d['roofline']['dominant_kernel']; print('$m $v #$rep: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s  serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$m $v failed', e)"
    done
  done
done
