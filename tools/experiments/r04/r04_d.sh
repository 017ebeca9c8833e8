# GPU box (round 4, call d): aligned task cuts alone (BSX_MULTI=0), then with k_hscan_multi
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04d; mkdir -p $O; cd $R
BSX_MULTI=0 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large or rrbs" > $O/pytest_nomulti.log 2>&1; echo "pytest (no multi) rc=$?"; tail -3 $O/pytest_nomulti.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large" > $O/pytest_multi_heavy.log 2>&1; echo "pytest (multi, heavy subset) rc=$?"; tail -3 $O/pytest_multi_heavy.log
for v in 1 0; do
  for m in pe; do
    BSX_MULTI=$v timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 > $O/bench_${m}_multi$v.json 2> $O/bench_${m}_multi$v.err
    python3 -c "
import json
try:
    d=json.load(open('$O/bench_${m}_multi$v.json')); k=d['roofline']['dominant_kernel']; print('$m multi=$v: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$m failed', e)"
  done
done
NOPYTEST= bash tools/quick.sh r04d_q "rrbs trim se"
