# GPU box (round 4, call b): RRBS parity with the plane form of k_hscan_shared, the oracle-built index check, gather microbenchmark, SQ passes of the plane k_hscan
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04b; mkdir -p $O; cd $R
$R/tools/microbench/gather_cost > $O/r04b_gather_cost.json 2> $O/gather.err; echo "gather rc=$?"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py tests/test_gpu_fullsize.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for rep in 1 2; do
timeout 600 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 3 > $O/bench_rrbs_$rep.json 2> $O/bench_rrbs_$rep.err
python3 -c "
import json
d=json.load(open('$O/bench_rrbs_$rep.json')); k=d['roofline']['dominant_kernel']; print('rrbs #$rep: %.1f ms/step  %.2f M reads/s   %s %.1f ms/step %.0f Gcand/s' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9))"
done
bash tools/sq_passes.sh r04b_sq; cat $R/gpurun_out/r04b_sq/r04b_sq_sq_derived.json | head -80
