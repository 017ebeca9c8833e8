# round 6, call 3: parity of the prefix replay; category clocks of the control kernel (C5) before / after; per-pass table; A/B lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06c; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "without_work_counters or heavy or context" > $O/parity.txt 2>&1; tail -n 3 $O/parity.txt
for v in ev0 default; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  for m in trim; do timeout 600 python3 tools/ctrl_clocks.py --mode $m > $O/ctrl_clocks_${m}_$v.json 2> $O/ctrl_clocks_${m}_$v.err; cut -c1-1200 $O/ctrl_clocks_${m}_$v.json; echo; done
done
unset BSX_LIB
for m in trim rrbs; do bash tools/pass_profile.sh r06c_new $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0; done
bash tools/ab_libs.sh r06c_ab ev0 "trim rrbs pe" 1
