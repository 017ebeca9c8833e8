# GPU box, round 5 (d): parity of the build with the work-counter switch (heavy subset + whole batches), then C3 by batch size, pools and batches in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large or rrbs or counters" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log
run() { # name, args...
  n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']; w=d['roofline'].get('with_work_counters') or {}
    print('$n: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s (counted %.1f ms) group_share %.3f serial %.1f (counted %.1f) pools %s' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, w.get('scan_kernel_ms_per_step') or 0, d['roofline']['group_share'], d['roofline']['serial_replay']['ms_per_step'], w.get('serial_ms_per_step') or 0, d['config']['heavy_pools'][0]))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-400:])"
}
run b1m_f3 --steps 12 --warmup 3
run b1m_f3_counted --steps 12 --warmup 3 --work-counters 1
run b2m_f3 --pairs-per-step 2097152 --steps 6 --warmup 3
run b4m_f2 --pairs-per-step 4194304 --in-flight 2 --steps 4 --warmup 2
run b4m_f2_p40k2m --pairs-per-step 4194304 --heavy-limits 40000,2097152 --in-flight 2 --steps 4 --warmup 2
run b4m_f3_p40k2m --pairs-per-step 4194304 --heavy-limits 40000,2097152 --in-flight 3 --steps 6 --warmup 3
run b4m_f2_p60k4m --pairs-per-step 4194304 --heavy-limits 60000,4194304 --in-flight 2 --steps 4 --warmup 2
run b8m_f2_p60k4m --pairs-per-step 8388608 --heavy-limits 60000,4194304 --in-flight 2 --steps 4 --warmup 2
