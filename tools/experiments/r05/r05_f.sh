# GPU box, round 5 (f): variants of k_hscan_same (uncounted) at the bench's default step size; end to end by CLI batch size
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
for v in default rowpf rowpf_w4 rowpf_nopf w4 c3_nopf c4_nopf_w4 default; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 4 --warmup 2 > $O/pe_$v.json 2> $O/pe_$v.err
  python3 -c "
import json
try:
    d=json.load(open('$O/pe_$v.json')); k=d['roofline']['dominant_kernel']; w=d['roofline'].get('with_work_counters') or {}
    print('$v: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s (counted %.1f ms) serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, w.get('scan_kernel_ms_per_step') or 0, d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$v failed', e)"
done
unset BSX_LIB
for cfg in "1050000 3" "2100000 3" "2100000 2" "4200000 2" "4200000 3"; do
  set -- $cfg
  BSX_BATCH=$1 BSX_GPU_BATCHES=$2 timeout 600 python3 tools/e2e_bench.py --pairs 16777216 > $O/e2e_$1_$2.json 2> $O/e2e_$1_$2.err
  python3 -c "
import json
try:
    d=json.load(open('$O/e2e_$1_$2.json')); t=d['timing']; print('e2e batch $1 x $2: %.2f M reads/s mapping %.2f s' % (d['reads_per_s_mapping_phase']/1e6, t['mapping_s']), t['stage_busy_s'])
except Exception as e: print('e2e $1 $2 failed', e)"
done
