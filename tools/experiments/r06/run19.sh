# round 6, call 19: smoke(); C4 with two batches in flight and full pools against the default three with two-thirds pools
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06s; mkdir -p $O; cd $R
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f  pools %s' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0, d['config']['heavy_pools'][0]))"; }
run() { tag=$1; shift; timeout 900 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run base_$rep --steps 9 --warmup 3
  run fl2_$rep --steps 8 --warmup 2 --in-flight 2
  run fl3_150k_$rep --steps 9 --warmup 3 --heavy-limits 150000,1800000
  run fl4_$rep --steps 12 --warmup 4 --in-flight 4 --heavy-limits 80000,1000000
done
