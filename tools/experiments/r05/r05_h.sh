# GPU box, round 5 (h): the new -m gpu tests, the bin-gather microbenchmark, the VALU mixes of k_hscan_same, compulsory sectors of the group scan
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_rccl.py -m gpu -x -q > $O/pytest_rccl.log 2>&1; echo "pytest rccl rc=$?"; tail -3 $O/pytest_rccl.log
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "capacity_limit or without_work_counters" > $O/pytest_limit.log 2>&1; echo "pytest limit rc=$?"; tail -3 $O/pytest_limit.log
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -x -q -k "bench_sized or index or whole_batch_equals_oracle_c3 or work_counters" > $O/pytest_full.log 2>&1; echo "pytest fullsize rc=$?"; tail -3 $O/pytest_full.log
cp gpurun_out/validate/*.json $O/ 2>/dev/null
timeout 600 tools/microbench/bin_gather > $O/bin_gather.json 2> $O/bin_gather.err; echo "bin_gather rc=$?"; cat $O/bin_gather.json
timeout 600 tools/microbench/valu_issue > $O/valu_issue.json 2> $O/valu_issue.err; echo "valu_issue rc=$?"; python3 -c "
import json
j=json.load(open('$O/valu_issue.json'))
for m in j['mixes'][-4:]:
    print(m['mix'][:90], {k:(round(v['simd_cycles_per_wave_instr'],2), round(v['chip_G_wave_instr_per_s'])) for k,v in m['by_waves_per_simd'].items()})"
BSX_LIB=$R/bsmap_amd/libbsx_sectors.so BSX_SECTOR_STATS=1 timeout 900 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 1 --warmup 0 --work-counters 1 2> $O/sectors.err > $O/sectors.json; grep sectors $O/sectors.err
BSX_LIB=$R/bsmap_amd/libbsx_sectors.so BSX_SECTOR_STATS=1 timeout 900 python3 bench.py --in-flight 1 --pairs-per-step 1048576 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 1 --warmup 0 --work-counters 1 2> $O/sectors_1m.err > $O/sectors_1m.json; grep sectors $O/sectors_1m.err
