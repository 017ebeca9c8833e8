# round 6, call 1: parity of the event-continue replay, per-pass tables before / after, default-mode lines before / after
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_work_counters or heavy" > $O/parity.txt 2>&1; tail -3 $O/parity.txt
for m in trim rrbs; do
  BSX_LIB=$R/bsmap_amd/libbsx_ev0.so bash tools/pass_profile.sh r06a_ev0 $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0
  bash tools/pass_profile.sh r06a_ev1 $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0
done
bash tools/ab_libs.sh r06a_ab ev0 "trim rrbs pe" 1
