# round 6, call 6: publish-ahead parity; A/B with repeats: default (publish ahead, chunks / 128) against no publish-ahead, chunks / 16 and the round-5 arrangement
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06f; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_work_counters or heavy or context" > $O/parity.txt 2>&1; tail -n 3 $O/parity.txt
for rep in 1 2; do for v in default nospec q16 r05; do for m in rrbs trim; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
  python3 -c "
import json
d=json.load(open('$O/${m}_${v}_$rep.json')); k=d['roofline']['dominant_kernel']; print('$m $v #$rep: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))"
done; done; done
unset BSX_LIB
for m in trim pe; do bash tools/pass_profile.sh r06f_new $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0; done
