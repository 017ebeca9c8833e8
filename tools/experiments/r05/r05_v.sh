# GPU box, round 5 (v): the context prefilter over 1 / 2 / 4 chunks per step (queue of survivors in LDS) against one chunk at a time with prefetch (head)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py -m gpu -x -q -k "without_work_counters" > $O/pytest_a.log 2>&1; echo "pytest counters-off suites (4 chunks per step) rc=$?"; tail -2 $O/pytest_a.log
for v in default head cw2 cw1 default head; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 2 > $O/pe_$v.json 2> $O/pe_$v.err
  python3 -c "
import json
try:
    d=json.load(open('$O/pe_$v.json')); k=d['roofline']['dominant_kernel']
    print('$v: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$v failed', e); print(open('$O/pe_$v.err').read()[-300:])"
done
