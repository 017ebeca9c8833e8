# round 6, call 18: where the tail begins: whole lists below 256 / windows of 2^24 below 2048 active units (default) against 512 / 4096, 1024 / 8192, 2048 / 16384
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06r; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0))"; }
for rep in 1 2; do for v in default tu4k tu8k tu16k; do for m in pe trim se; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
  line $O/${m}_${v}_$rep.json "$m $v #$rep"
done; done; done
