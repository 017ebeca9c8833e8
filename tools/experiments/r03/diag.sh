# GPU box: design evidence for the device-driven heavy pipeline (round 3).  usage: bash tools/experiments/r03/diag.sh  -> gpurun_out/r03diag/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03diag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/microbench/launch_cost > $O/launch_cost.json 2> $O/launch_cost.err; echo "launch_cost rc=$?"; cat $O/launch_cost.json
for m in pe rrbs trim; do
  for mode in serial default; do
    S=/tmp/tl_${m}_$mode; rm -rf $S
    if [ $mode = serial ]; then args="--profile-serial --steps 2 --warmup 1"; else args="--cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --steps 4 --warmup 2 --in-flight 1"; fi
    rocprofv3 --kernel-trace -d $S -o t --output-format csv -- python3 $R/bench.py --mode $m $args > $O/bench_${m}_$mode.json 2> $O/bench_${m}_$mode.err; echo "$m $mode rc=$?"
    python3 $R/tools/timeline.py $S > $O/timeline_${m}_$mode.json; cat $O/timeline_${m}_$mode.json | cut -c1-1500
    rm -rf $S
  done
  BSX_TRACE_HEAVY=1 python3 $R/tools/ctrl_clocks.py --mode $m > $O/ctrl_clocks_$m.json 2> $O/ctrl_trace_$m.txt; echo "clocks $m rc=$?"; cat $O/ctrl_clocks_$m.json
done
