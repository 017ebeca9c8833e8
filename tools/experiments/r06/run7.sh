# round 6, call 7: tail windows (2^24 below 2048 active units, 2^27 below 256) against 2^22; RRBS deferral threshold; parity of the heavy cases
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06g; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heavy" > $O/parity.txt 2>&1; tail -n 3 $O/parity.txt
line() { python3 -c "
import json
d=json.load(open('$1')); k=d['roofline']['dominant_kernel']; print('$2: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))"; }
for rep in 1 2; do for v in default wt22; do for m in trim pe se; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
  line $O/${m}_${v}_$rep.json "$m $v #$rep"
done; done; done
unset BSX_LIB
for t in 1024 2048 4096 8192 16384; do
  timeout 600 python3 bench.py --mode rrbs --heavy-threshold $t --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/rrbs_t$t.json 2> $O/rrbs_t$t.err
  line $O/rrbs_t$t.json "rrbs threshold $t"
done
for m in trim pe; do bash tools/pass_profile.sh r06g_new $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0; done
