# GPU box: parity subset (golden / synthetic / whole-batch) + one default-mode bench line per mode.  usage: bash tools/quick.sh <tag> [modes]
TAG=${1:-q}; MODES=${2:-"pe se rrbs trim"}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
[ -n "$NOPYTEST" ] || { timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log; }
for m in $MODES; do
  for rep in 1 2; do
    timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 > $O/bench_${m}_$rep.json 2> $O/bench_${m}_$rep.err
    python3 -c "
import json
try:
    d=json.load(open('$O/bench_${m}_$rep.json')); k=d['roofline']['dominant_kernel']; print('$m #$rep: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9))
except Exception as e: print('$m failed', e)"
  done
done
