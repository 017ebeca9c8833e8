# round 6, call 21: C5 with a larger task pool (its early passes ask for 2.4 M tasks of a 1.57 M pool)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06u; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f  pools %s' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0, d['config']['heavy_pools'][0]))"; }
run() { tag=$1; shift; timeout 900 python3 bench.py --mode trim --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run base_$rep
  run t25_$rep --heavy-limits 31457,2500000
  run t35_$rep --heavy-limits 31457,3500000
  run u40_t25_$rep --heavy-limits 40000,2500000
done
