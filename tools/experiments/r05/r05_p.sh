# GPU box, round 5 (p): the final bench lines (three batches in flight for every mode): default, the driver's command, the other modes with their legs
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O; cd $R
timeout 1500 python3 bench.py > $O/r05p_bench.json 2> $O/r05p_bench.err; echo "default rc=$?"
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05p_bench_driver_command.json 2> $O/r05p_bench_driver_command.err; echo "driver's command rc=$?"
for m in se rrbs trim; do timeout 900 python3 bench.py --mode $m --e2e-pairs 0 > $O/r05p_bench_$m.json 2> $O/r05p_bench_$m.err; echo "$m rc=$?"; done
timeout 900 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 > $O/r05p_bench_one_in_flight.json 2> /dev/null; echo "one in flight rc=$?"
timeout 900 python3 bench.py --in-flight 2 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 > $O/r05p_bench_two_in_flight.json 2> /dev/null; echo "two in flight rc=$?"
python3 -c "
import json
for n in ('bench','bench_driver_command','bench_se','bench_rrbs','bench_trim','bench_one_in_flight','bench_two_in_flight'):
    try:
        j=json.loads([l for l in open('$O/r05p_%s.json' % n) if l.startswith('{')][-1]); r=j['roofline']
        print(n, round(j['value']/1e6,2), round(j['ms_per_step'],1), 'frac', r['frac'], 'fabric', (r.get('fabric_requests') or {}).get('frac'), 'incl', (j.get('value_incl_transfers') or {}).get('value'), {k:round(v.get('reads_per_s',0)/1e6,2) for k,v in (j.get('other_configs') or {}).items()}, 'cpu', (j.get('cpu_baseline') or {}).get('value'), 'e2e', (j.get('end_to_end') or {}).get('reads_per_s'), 'scan', round(r['dominant_kernel']['ms_per_step'],1), 'serial', (r.get('serial_replay') or {}).get('ms_per_step'))
    except Exception as e: print(n, 'failed', e)
"
