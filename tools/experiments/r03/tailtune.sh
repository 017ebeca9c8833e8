# GPU box: tail-mode threshold / grid.  usage: bash tools/experiments/r03/tailtune.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for cfg in "16384 4096" "65536 4096" "65536 16384" "262144 16384" "1048576 65536"; do
  set -- $cfg
  for nf in 2 1; do
    BSX_TAIL_TASKS=$1 BSX_TAIL_GRID=$2 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 --transfer-steps 0 --in-flight $nf > $O/b_$1_$2_f$nf.json 2> $O/b_$1_$2_f$nf.err
    python3 -c "
import json; d=json.load(open('$O/b_$1_$2_f$nf.json')); print('tail_tasks $1 grid $2 in flight $nf: %.1f ms/step %.2f M' % (d['ms_per_step'], d['value']/1e6))"
  done
done
for cfg in "16384 4096" "262144 16384"; do
  set -- $cfg
  for m in trim rrbs; do
    BSX_TAIL_TASKS=$1 BSX_TAIL_GRID=$2 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/b_$1_$2_$m.json 2> $O/b_$1_$2_$m.err
    python3 -c "
import json; d=json.load(open('$O/b_$1_$2_$m.json')); print('tail_tasks $1 grid $2 $m: %.1f ms/step %.2f M' % (d['ms_per_step'], d['value']/1e6))"
  done
done
