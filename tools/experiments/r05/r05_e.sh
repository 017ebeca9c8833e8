# GPU box, round 5 (e): parity subset again, then the other modes by batch size / batches in flight, and the default line with the driver's arguments
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large or rrbs or counters" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
run() { # name, args...
  n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']; w=d['roofline'].get('with_work_counters') or {}
    print('$n: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s (counted %.1f ms) group_share %.3f serial %.1f (counted %.1f) pools %s heavy %d' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, w.get('scan_kernel_ms_per_step') or 0, d['roofline']['group_share'], d['roofline']['serial_replay']['ms_per_step'], w.get('serial_ms_per_step') or 0, d['config']['heavy_pools'][0], d['roofline']['heavy_units_last_step']))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-400:])"
}
run pe_default --steps 8 --warmup 2
run se_1m_f3 --mode se --pairs-per-step 1048576 --in-flight 3 --steps 6 --warmup 3
run se_4m_f2 --mode se --steps 4 --warmup 2
run se_2m_f3 --mode se --pairs-per-step 2097152 --in-flight 3 --steps 6 --warmup 3
run trim_1m_f3 --mode trim --steps 6 --warmup 3
run trim_2m_f2 --mode trim --pairs-per-step 2097152 --in-flight 2 --steps 4 --warmup 2
run trim_4m_f2 --mode trim --pairs-per-step 4194304 --in-flight 2 --steps 4 --warmup 2
run rrbs_1m_f3 --mode rrbs --steps 6 --warmup 3
run rrbs_2m_f2 --mode rrbs --pairs-per-step 2097152 --in-flight 2 --steps 4 --warmup 2
run rrbs_2m_f3 --mode rrbs --pairs-per-step 2097152 --in-flight 3 --heavy-limits 110000,1400000 --steps 6 --warmup 3
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/drv.json 2> $O/drv.err; echo "driver command rc=$?"
python3 -c "
import json
j=json.loads([l for l in open('$O/drv.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step']); print(json.dumps(j.get('value_incl_transfers'))[:400]); print(json.dumps(j.get('other_configs'))[:1800]); print(json.dumps(j['config'])[:1500]); print(json.dumps(j['roofline'].get('serial_replay'))); print(json.dumps(j['roofline'].get('with_work_counters'))); print(json.dumps(j.get('end_to_end'))[:300]); print(json.dumps(j.get('cpu_baseline'))[:300]); print(json.dumps(j.get('sensitivity'))[:600])"
