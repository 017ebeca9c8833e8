# round 6, call 20: lone tasks as groups of one (-DBSX_LONE_GROUP=1) against hp_task; C4 with the larger default pools
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06t; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f  pools %s' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0, d['config']['heavy_pools'][0]))"; }
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heavy_pipeline_large or heavy_pipeline_without or runs_its_scan" > $O/parity_default.txt 2>&1; tail -n 2 $O/parity_default.txt
BSX_LIB=$R/bsmap_amd/libbsx_lone1.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heavy_pipeline_large or heavy_pipeline_without" > $O/parity_lone1.txt 2>&1; tail -n 2 $O/parity_lone1.txt
for rep in 1 2; do for v in default lone1; do for m in pe se trim; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
  line $O/${m}_${v}_$rep.json "$m $v #$rep"
done; done; done
unset BSX_LIB
for rep in 1 2; do timeout 600 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/rrbs_$rep.json 2> $O/rrbs_$rep.err; line $O/rrbs_$rep.json "rrbs default pools #$rep"; done
