# GPU box, round 5 (s): the context prefilter of the main kernel (16 bytes of flanking reference per index entry): parity where the counters are off, then A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py -m gpu -x -q -k "without_work_counters or golden or random_option or edge" > $O/pytest_a.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_a.log
BSX_WORK_COUNTERS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or random_option or edge or empty or heavy_pipeline_large" > $O/pytest_b.log 2>&1; echo "pytest parity, counters off everywhere rc=$?"; tail -3 $O/pytest_b.log
timeout 1500 python3 -m pytest tests/test_gpu_cli.py -m gpu -x -q > $O/pytest_cli.log 2>&1; echo "pytest cli rc=$?"; tail -3 $O/pytest_cli.log
run() { n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']
    print('$n: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f (counted %.1f) pools %s' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], (d['roofline'].get('serial_replay') or {}).get('ms_per_step') or 0, (d['roofline'].get('with_work_counters') or {}).get('serial_ms_per_step') or 0, d['config']['heavy_pools']))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-300:])"
}
run ctx_f2 --steps 6 --warmup 2 --in-flight 2
BSX_CTX=0 run noctx_f2 --steps 6 --warmup 2 --in-flight 2
run ctx_f3 --steps 6 --warmup 3 --in-flight 3
run ctx_f2_again --steps 6 --warmup 2 --in-flight 2
run se_ctx_f3 --mode se --steps 6 --warmup 3
BSX_CTX=0 run se_noctx_f3 --mode se --steps 6 --warmup 3
