# round 6, call 8: the command line end to end at hg38 size on one GPU: one pipeline (batch sizes) against lanes that share the GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06h; mkdir -p $O; cd $R
free -g | head -2; df -h /dev/shm | tail -1; nproc; cat /sys/fs/cgroup/cpu.max
V=";BSX_BATCH=2100000 BSX_GPU_BATCHES=2;BSX_BATCH=2100000 BSX_GPU_BATCHES=3;--lanes=2 --lane-files;--lanes=3 --lane-files;--lanes=4 --lane-files;--lanes=2"
timeout 2400 python3 tools/e2e_bench.py --pairs 16777216 --dir /dev/shm/bsx_e2e_$$ --variants "$V" > $O/e2e.json 2> $O/e2e.err; echo rc=$?; tail -n 5 $O/e2e.err
python3 -c "
import json
d=json.load(open('$O/e2e.json'))
for r in d.get('variants', [d]):
    t=r['timing']; n=2*r['pairs']; c=t['mapping_cpu_s']
    print('[%-44s] mapping %.2f s = %5.1f M reads/s | whole %.2f s = %5.1f M | cpu %.1f s | join %.2f s | load %.2f idx %.2f' % (r.get('cli_args',''), t['mapping_s'], n/t['mapping_s']/1e6, r['cli_wall_s'], n/r['cli_wall_s']/1e6, c['user']+c['sys'], t.get('join_s', 0.0), t.get('load_reference_s',0), t.get('index_s',0)))"
line() { python3 -c "
import json
d=json.load(open('$1')); k=d['roofline']['dominant_kernel']; print('$2: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))"; }
for rep in 1 2; do for v in default r05; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 > $O/rrbs_${v}_$rep.json 2> $O/rrbs_${v}_$rep.err
  line $O/rrbs_${v}_$rep.json "rrbs $v #$rep"
done; done
