# GPU box: the round's artefacts for the CURRENT build, summarised on the box (raw traces are too big to travel back).
# usage: bash tools/profile_round.sh <tag>   ->  gpurun_out/<tag>/  (copy what is to be kept into profiles/)
#   <tag>_bench*.json                 default bench line (C3, all legs) and the C2 / C4 / C5 lines
#   <tag>_kernel_stats.csv            rocprofv3 --kernel-trace --stats of the serial-mode bench (no two kernels overlap)
#   <tag>_default_kernel_stats.csv    the same of the default (two batches in flight) bench
#   <tag>_pmc.json, <tag>_sq.json     FETCH_SIZE / WRITE_SIZE and SQ / TA / TCC counter passes (serial mode, separate runs)
#   <tag>_valu_issue.json             VALU issue microbenchmark
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export BSX_PROFILES_DIR=$O
export BSX_PROFILE_MODE=pe BSX_PROFILE_WORK_COUNTERS=0 BSX_PROFILE_STEPS=3   # (the counter passes below: C3, work counters off as the timed region runs, --steps 2 --warmup 1)
S=/tmp/bsx_prof_$$; mkdir -p $S
cd $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $S/stats -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/${TAG}_bench_serial.json 2> $S/stats.log; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats -d $S/stats_default -o s --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 4 --warmup 2 > $O/${TAG}_bench_under_rocprof.json 2> $S/stats_default.log; echo "stats default rc=$?"
cut -c1-160 $S/stats_default/s_kernel_stats.csv > $O/${TAG}_default_kernel_stats.csv
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $S/pmc_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> $S/pmc_$t.log; echo "$t rc=$?"
done
python3 $R/tools/summarize_pmc.py $TAG $S/stats $S/pmc_FETCH_SIZE $S/pmc_WRITE_SIZE > /dev/null; echo "summarize_pmc rc=$?"
python3 $R/tools/summarize_sq.py $TAG $S/pmc_SQ_WAVES $S/pmc_SQ_WAIT_INST_ANY $S/pmc_TA_TA_BUSY_sum $S/pmc_SQ_LDS_BANK_CONFLICT > /dev/null; echo "summarize_sq rc=$?"
cp $O/${TAG}_sq.json $R/profiles/${TAG}_sq.json  # (on the box) the bench lines below quote this build's counter summary
cp $O/pmc_latest.json $R/profiles/pmc_latest.json  # (on the box) so that the bench lines below carry this build's traffic
# the other modes' scan kernels (C2: k_hscan_same on 100-nt reads; C4: k_hscan_shared; C5): their own counter summaries, so that their bench lines quote their own fractions
for m in se rrbs trim; do
  export BSX_PROFILE_MODE=$m
  DIRS=""
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    t=$(echo $set | cut -d' ' -f1)
    rocprofv3 --pmc $set --kernel-trace -d $S/${m}_$t -o p --output-format csv -- python3 $R/bench.py --mode $m --profile-serial --steps 2 --warmup 1 > /dev/null 2> $S/${m}_$t.log; echo "$m $t rc=$?"
    DIRS="$DIRS $S/${m}_$t"
  done
  python3 $R/tools/summarize_sq.py ${TAG}_$m $DIRS > /dev/null; echo "summarize_sq $m rc=$?"
  mv $O/${TAG}_${m}_sq.json $O/${TAG}_sq_$m.json; cp $O/${TAG}_sq_$m.json $R/profiles/${TAG}_sq_$m.json
done
export BSX_PROFILE_MODE=pe
cd $R
timeout 1200 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "default rc=$?"
for m in se rrbs trim; do
  timeout 900 python3 bench.py --mode $m --e2e-pairs 0 > $O/${TAG}_bench_$m.json 2> $O/${TAG}_bench_$m.err; echo "$m rc=$?"
done
timeout 900 python3 bench.py --mode trim --in-flight 2 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/${TAG}_bench_trim_f2.json 2> $O/${TAG}_bench_trim_f2.err; echo "trim f2 rc=$?"
timeout 900 python3 bench.py --mode trim --exact --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/${TAG}_bench_trim_exact.json 2> $O/${TAG}_bench_trim_exact.err; echo "trim exact rc=$?"
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/${TAG}_bench_driver_command.json 2> $O/${TAG}_bench_driver_command.err; echo "driver's command rc=$?"
timeout 900 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 > $O/${TAG}_bench_one_in_flight.json 2> $O/${TAG}_bench_one_in_flight.err; echo "one in flight rc=$?"
[ -n "$HOST" ] && { bash $R/tools/host_ceiling.sh ${TAG} > $O/${TAG}_host_ceiling.txt 2>&1; cp $R/gpurun_out/${TAG}_hc_*.json $O/ 2>/dev/null; cat $O/${TAG}_host_ceiling.txt; }   # (HOST=1: the host-side legs, unchanged since round 4)
$R/tools/microbench/valu_issue > $O/${TAG}_valu_issue.json 2>/dev/null; echo "valu rc=$?"
$R/tools/microbench/gather_cost > $O/${TAG}_gather_cost.json 2>/dev/null; echo "gather rc=$?"
[ -n "$HOST" ] && { bash $R/tools/lanes_scaling.sh ${TAG}_lanes > $O/${TAG}_lanes.txt 2>&1; cp $R/gpurun_out/${TAG}_lanes/*.json $O/ 2>/dev/null; cat $O/${TAG}_lanes.txt; }
rm -rf $S; ls -la $O | head -30
timeout 1500 python3 tools/validate_fullsize.py --mode pe > $O/${TAG}_validate_full_c3.json 2>/dev/null; echo "validate full rc=$?"
# A/B on this box: the group scan kernel (k_hscan_same, the WGBS default) against the one-task kernel (BSX_SAME=0), paired and single-end; RRBS through
# k_hscan_shared (its default) and through k_hscan_same (BSX_SAME=2); and the histogram of tasks per identical window / window and read offset
for m in pe se rrbs; do for v in 1 0 2; do
  if [ $m = rrbs ]; then [ $v = 0 ] && continue; else [ $v = 2 ] && continue; fi
  BSX_SAME=$v timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 8 --warmup 2 > $O/${TAG}_ab_${m}_same$v.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/${TAG}_ab_${m}_same$v.json')); k=d['roofline']['dominant_kernel']; print('$m BSX_SAME=$v: %.1f ms/step %.2f M reads/s | %s %.1f ms/step %.0f G candidates/s' % (d['ms_per_step'], d['value']/1e6, k['name'], k['ms_per_step'], k['candidates_per_s']/1e9))"
done; done
BSX_SIGHIST=1 timeout 600 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 2 --warmup 1 2>&1 >/dev/null | grep sighist > $O/${TAG}_sighist.txt; head -3 $O/${TAG}_sighist.txt
