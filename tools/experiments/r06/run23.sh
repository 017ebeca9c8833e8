# round 6, call 23: C4 (RRBS) through the group scan kernel (BSX_SAME=2) now that its control passes are shorter (round 5: the scan won 16 %, the step nothing)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0))"; }
run() { tag=$1; shift; timeout 900 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run shared_$rep
  BSX_SAME=2 run same_$rep
done
BSX_SAME=2 run same_fl2 --in-flight 2
BSX_SAME=2 HG_DUMMY=1 run same_t2m --heavy-limits 160000,2500000
