# GPU box, round 5 (c): does a larger device batch pay?  More deferred units per pass = more tasks over the same window AND offset = larger groups
# for k_hscan_same (fewer lone tasks).  reads/s of C3 by units per step and pools.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O; cd $R
run() { # name, args...
  n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']; print('$n: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s  group_share %.3f serial %.1f pools %s' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, d['roofline']['group_share'], d['roofline']['serial_replay']['ms_per_step'], d['config']['heavy_pools'][0]))
except Exception as e: print('$n failed', e); import subprocess; print(open('$O/$n.err').read()[-400:])"
}
run b1m_f3 --steps 12 --warmup 3
run b2m_f2 --pairs-per-step 2097152 --heavy-limits 55638,2097152 --in-flight 2 --steps 6 --warmup 2
run b2m_f3 --pairs-per-step 2097152 --heavy-limits 55638,2097152 --in-flight 3 --steps 6 --warmup 3
run b4m_f2 --pairs-per-step 4194304 --heavy-limits 60000,4194304 --in-flight 2 --steps 4 --warmup 2
run b4m_f1 --pairs-per-step 4194304 --heavy-limits 60000,4194304 --in-flight 1 --steps 3 --warmup 1
run b2m_f2_dflt --pairs-per-step 2097152 --in-flight 2 --steps 6 --warmup 2
