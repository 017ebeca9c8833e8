# round 6, call 9: single-end control kernel capped at 168 / 128 VGPRs (more scan waves beside it) against 224; lanes variants end to end
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06i; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); k=d['roofline']['dominant_kernel']; print('$2: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))"; }
for rep in 1 2; do for v in default hw3 hw4; do for m in rrbs se; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/${m}_${v}_$rep.json 2> $O/${m}_${v}_$rep.err
  line $O/${m}_${v}_$rep.json "$m $v #$rep"
done; done; done
unset BSX_LIB
V="--lanes=2 --lane-files;--lanes=2 --lane-files BSX_BATCH=1050000;--lanes=2 --lane-files BSX_BATCH=2100000 BSX_GPU_BATCHES=2;--lanes=2 --lane-files BSX_GPU_BATCHES=2;--lanes=2 --lane-files -p 5;--lanes=2 --lane-files -p 7"
timeout 2400 python3 tools/e2e_bench.py --pairs 16777216 --dir /dev/shm/bsx_e2e_$$ --variants "$V" > $O/e2e.json 2> $O/e2e.err; echo rc=$?; tail -n 5 $O/e2e.err
python3 -c "
import json
d=json.load(open('$O/e2e.json'))
for r in d.get('variants', [d]):
    t=r['timing']; n=2*r['pairs']; c=t['mapping_cpu_s']
    print('[%-60s] mapping %.2f s = %5.1f M reads/s | whole %.2f s | cpu %.1f s | load %.2f' % (r.get('cli_args',''), t['mapping_s'], n/t['mapping_s']/1e6, r['cli_wall_s'], c['user']+c['sys'], t.get('load_reference_s',0)))"
