# GPU box, round 5 (q): the coming step's reference pairs through LDS-DMA (global_load_lds_dwordx4) in k_hscan_same: parity of the variant, then A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O; cd $R
BSX_LIB=$R/bsmap_amd/libbsx_glds3.so timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "heavy_pipeline_large or without_work_counters or scan_grids" > $O/pytest_glds.log 2>&1; echo "pytest (glds variant) rc=$?"; tail -3 $O/pytest_glds.log
for v in default glds3 glds3_c3 glds_c2 default; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 > $O/pe_$v.json 2> $O/pe_$v.err
  python3 -c "
import json
try:
    d=json.load(open('$O/pe_$v.json')); k=d['roofline']['dominant_kernel']
    print('$v: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9, d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$v failed', e); print(open('$O/pe_$v.err').read()[-300:])"
done
