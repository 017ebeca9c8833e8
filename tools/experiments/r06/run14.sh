# round 6, call 14: C5 with two control blocks per CU, and with four batches in flight (smaller pools)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06n; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0))"; }
run() { tag=$1; shift; timeout 900 python3 bench.py --mode trim --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run base_$rep --steps 9 --warmup 3
  BSX_HCTRL_BLOCKS=2 run hb2_$rep --steps 9 --warmup 3
  run fl4_$rep --steps 12 --warmup 4 --in-flight 4 --heavy-limits 20000,1000000
  run fl4b_$rep --steps 12 --warmup 4 --in-flight 4 --pairs-per-step 2097152 --heavy-limits 20000,1000000
  run fl2big_$rep --steps 8 --warmup 2 --in-flight 2 --pairs-per-step 4194304
done
