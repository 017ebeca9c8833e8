# round 6, call 22: the main kernel's waiting ring (BSX_DENSE: candidates behind the context prefilter gathered 64 at a time) against -DBSX_DENSE=0, C3 and C2; then parity
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06x; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0))"; }
run() { tag=$1; lib=$2; mode=$3; shift 3; BSX_LIB=$lib timeout 900 python3 bench.py --mode $mode --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 8 --warmup 2 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run pe_dense_$rep bsmap_amd/libbsx.so pe
  run pe_nodense_$rep bsmap_amd/libbsx_nodense.so pe
done
run se_dense bsmap_amd/libbsx.so se
run se_nodense bsmap_amd/libbsx_nodense.so se
run trim_dense bsmap_amd/libbsx.so trim
run trim_nodense bsmap_amd/libbsx_nodense.so trim
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_leak_exact.py tests/test_gpu_parity.py -x -q -m gpu -k "fullsize or leak_exact or without_work_counters" > $O/tests.txt 2>&1; tail -n 4 $O/tests.txt
