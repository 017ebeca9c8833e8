# GPU box, round 5 (i): segments (a wave takes all the sub-groups of a 64-slot segment over one window): parity, then A/B against one group per wave
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05j; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy or large or rrbs or counters" > $O/pytest_parity.log 2>&1; echo "pytest parity rc=$?"; tail -3 $O/pytest_parity.log
run() { # name, args...
  n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']; w=d['roofline'].get('with_work_counters') or {}
    print('$n: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s (counted %.1f ms) group_share %.3f serial %.1f (counted %.1f)' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, w.get('scan_kernel_ms_per_step') or 0, d['roofline']['group_share'], d['roofline']['serial_replay']['ms_per_step'], w.get('serial_ms_per_step') or 0))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-400:])"
}
run seg1 --steps 4 --warmup 2
BSX_SEG=0 run seg0 --steps 4 --warmup 2
BSX_LIB=$R/bsmap_amd/libbsx_entpf.so run seg1_entpf --steps 4 --warmup 2
run seg1_again --steps 4 --warmup 2
run se_seg1 --mode se --steps 4 --warmup 2
BSX_SEG=0 run se_seg0 --mode se --steps 4 --warmup 2
BSX_LIB=$R/bsmap_amd/libbsx_sectors.so BSX_SECTOR_STATS=1 timeout 900 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 1 --warmup 0 --work-counters 1 2> $O/sectors.err > $O/sectors.json; grep sectors $O/sectors.err
BSX_SIGHIST=1 timeout 600 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 2 --warmup 1 --work-counters 1 2>&1 >/dev/null | grep "sighist. scan"
