R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for la in 0 1; do
  S=/tmp/la_$la; rm -rf $S
  BSX_LOOKAHEAD=$la BSX_TRACE_HEAVY=1 rocprofv3 --kernel-trace -d $S -o t --output-format csv -- python3 $R/bench.py --mode rrbs --profile-serial --steps 2 --warmup 1 > $O/serial_la$la.json 2> $O/serial_la$la.err
  python3 $R/tools/timeline.py $S | cut -c1-700
  grep "bsx heavy" $O/serial_la$la.err | tail -40 | awk '{print $10, $12, $14}' | tr '\n' ';' | cut -c1-900; echo
done
