# GPU box, round 5 (t): the context prefilter: the counters-off suites, whole batches, then trim / se configurations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05t; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py -m gpu -x -q -k "without_work_counters" > $O/pytest_a.log 2>&1; echo "pytest counters-off suites rc=$?"; tail -3 $O/pytest_a.log
BSX_WORK_COUNTERS=0 timeout 1500 python3 -m pytest tests/test_gpu_synth.py -m gpu -x -q > $O/pytest_synth.log 2>&1; echo "pytest synth, counters off rc=$?"; tail -2 $O/pytest_synth.log
timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log
run() { n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']
    print('$n: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], (d['roofline'].get('serial_replay') or {}).get('ms_per_step') or 0))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-300:])"
}
run trim_4m_f2 --mode trim --steps 4 --warmup 2
run trim_3m_f3 --mode trim --pairs-per-step 3145728 --in-flight 3 --steps 6 --warmup 3
run trim_2m_f3 --mode trim --pairs-per-step 2097152 --in-flight 3 --steps 6 --warmup 3
run se_4m_f2 --mode se --in-flight 2 --steps 4 --warmup 2
run pe_6m_f2 --pairs-per-step 6291456 --steps 4 --warmup 2
