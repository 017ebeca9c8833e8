# round 6, call 10: exact mode with the context prefilter (tests, cost on C5); control clocks with the overflow-rest category; shmem huge pages?
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06j; mkdir -p $O; cd $R
cat /sys/kernel/mm/transparent_hugepage/shmem_enabled /sys/kernel/mm/transparent_hugepage/enabled 2>&1 | head -3
timeout 1500 python3 -m pytest tests/test_gpu_leak_exact.py tests/test_gpu_cli.py -x -q -m gpu -k "exact" > $O/exact.txt 2>&1; tail -n 3 $O/exact.txt
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f  binding %s' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0, r['binding_kernel']))"; }
for rep in 1 2; do for x in "" "--exact"; do
  timeout 600 python3 bench.py --mode trim $x --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/trim_x${x}_$rep.json 2> $O/trim_x${x}_$rep.err
  line $O/trim_x${x}_$rep.json "trim [$x] #$rep"
done; done
for m in trim pe rrbs; do timeout 600 python3 tools/ctrl_clocks.py --mode $m > $O/ctrl_clocks_${m}.json 2> $O/ctrl_clocks_${m}.err; cut -c1-1300 $O/ctrl_clocks_${m}.json; echo; done
