R=$GRAFT_REPO_ROOT; cd $R
for lib in default nt; do
  if [ $lib = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$lib.so; fi
  for w in 0 12 8 4; do
    for nfl in 2 3; do
      timeout 300 python3 bench.py --steps 6 --warmup 2 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --waves-per-cu $w --in-flight $nfl > /tmp/o.json 2>/dev/null
      python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$lib waves/cu=$w in-flight=$nfl: %.1f ms/step' % d['ms_per_step'])"
    done
  done
done
