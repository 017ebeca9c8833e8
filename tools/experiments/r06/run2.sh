# round 6, call 2: category clocks of the control kernel (C5, C4) before / after; parity of the prefix replay; per-pass tables; two control blocks per CU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_work_counters or heavy or context" > $O/parity.txt 2>&1; tail -n 3 $O/parity.txt
for v in ev0 default; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  for m in trim rrbs; do timeout 600 python3 tools/ctrl_clocks.py --mode $m > $O/ctrl_clocks_${m}_$v.json 2> $O/ctrl_clocks_${m}_$v.err; cut -c1-1200 $O/ctrl_clocks_${m}_$v.json; echo; done
done
unset BSX_LIB
for m in trim rrbs; do bash tools/pass_profile.sh r06b_new $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0; done
for hb in 1 2; do for m in trim rrbs pe; do
  BSX_HCTRL_BLOCKS=$hb timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 8 --warmup 2 > $O/${m}_hb$hb.json 2> $O/${m}_hb$hb.err
  python3 -c "
import json
d=json.load(open('$O/${m}_hb$hb.json')); k=d['roofline']['dominant_kernel']; print('$m hctrl_blocks $hb: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))"
done; done
