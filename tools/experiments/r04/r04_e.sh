# GPU box (round 4, call e): k_hscan_multi with pipelined staging: heavy parity subset, bench A/B against one task per wave, share of candidates in runs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04e; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large" > $O/pytest_multi_heavy.log 2>&1; echo "pytest (multi, heavy subset) rc=$?"; tail -3 $O/pytest_multi_heavy.log
for v in 1 0; do
  for m in pe; do
    BSX_MULTI=$v timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 > $O/bench_${m}_multi$v.json 2> $O/bench_${m}_multi$v.err
    python3 -c "
import json
try:
    d=json.load(open('$O/bench_${m}_multi$v.json')); k=d['roofline']['dominant_kernel']; print('$m multi=$v: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, d['roofline']['serial_replay']['ms_per_step']), d['roofline'].get('multi_share'))
except Exception as e: print('$m failed', e)"
  done
done
