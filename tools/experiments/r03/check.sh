# GPU box: parity subset + quick bench lines of the current build.  usage: bash tools/experiments/r03/check.sh <tag> [groups list] [modes]
TAG=${1:-chk}; GROUPS_LIST=${2:-"2 4"}; MODES=${3:-"pe rrbs trim"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
[ -n "$NOPYTEST" ] || timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for m in $MODES; do
  for g in $GROUPS_LIST; do
    for nf in 1 2; do
      BSX_HEAVY_GROUPS=$g timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --steps 6 --warmup 2 --in-flight $nf > $O/bench_${m}_g${g}_f$nf.json 2> $O/bench_${m}_g${g}_f$nf.err
      python3 -c "
import json
try:
    d=json.load(open('$O/bench_${m}_g${g}_f$nf.json')); print('$m groups $g in-flight $nf: %.1f ms/step  %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))
except Exception as e: print('$m g$g f$nf failed', e)"
    done
  done
done
