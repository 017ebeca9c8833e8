# GPU box, round 5 (x): the context prefilter for single-end RRBS lists in the main kernel: parity with the counters off, then C4 by deferral threshold
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
BSX_WORK_COUNTERS=0 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "rrbs" > $O/pytest_rrbs.log 2>&1; echo "pytest rrbs, counters off rc=$?"; tail -3 $O/pytest_rrbs.log
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "c4" > $O/pytest_c4.log 2>&1; echo "pytest fullsize c4 rc=$?"; tail -2 $O/pytest_c4.log
timeout 900 python3 -m pytest tests/test_gpu_cli.py -m gpu -x -q -k "rrbs or c4" > $O/pytest_cli.log 2>&1; echo "pytest cli rrbs rc=$?"; tail -2 $O/pytest_cli.log
run() { n=$1; shift
  timeout 900 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']
    print('$n: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f heavy %d' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], (d['roofline'].get('serial_replay') or {}).get('ms_per_step') or 0, d['roofline']['heavy_units_last_step']))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-300:])"
}
run ctx_t2048
BSX_CTX=0 run noctx_t2048
run ctx_t8192 --heavy-threshold 8192
run ctx_t32768 --heavy-threshold 32768
run ctx_t131072 --heavy-threshold 131072
