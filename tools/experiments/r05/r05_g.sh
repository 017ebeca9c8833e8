# GPU box, round 5 (g): chunks per step / waves per SIMD of k_hscan_same (uncounted), C3 at the default step size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O; cd $R
for v in ${VARIANTS:-default}; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode ${MODE:-pe} --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 4 --warmup 2 > $O/${MODE:-pe}_$v.json 2> $O/${MODE:-pe}_$v.err
  python3 -c "
import json
try:
    d=json.load(open('$O/${MODE:-pe}_$v.json')); k=d['roofline']['dominant_kernel']; w=d['roofline'].get('with_work_counters') or {}
    print('$v: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s (counted %.1f ms) serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, w.get('scan_kernel_ms_per_step') or 0, d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$v failed', e)"
done
