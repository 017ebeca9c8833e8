# GPU box, round 5 (n): the bench lines again with the final tree (same library as the r05m counter files): default, the driver's command, and the launcher at N = 1
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O; cd $R
timeout 1500 python3 bench.py > $O/r05n_bench.json 2> $O/r05n_bench.err; echo "default rc=$?"
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05n_bench_driver_command.json 2> $O/r05n_bench_driver_command.err; echo "driver's command rc=$?"
python3 -c "
import json
for n in ('bench','bench_driver_command'):
    j=json.loads([l for l in open('$O/r05n_%s.json' % n) if l.startswith('{')][-1]); r=j['roofline']
    print(n, round(j['value']/1e6,2), round(j['ms_per_step'],1), 'frac', r['frac'], 'fabric', r['fabric_requests']['frac'], r['fabric_requests']['requests_per_step'], 'incl', j['value_incl_transfers'].get('value'), {k:round(v.get('reads_per_s',0)/1e6,2) for k,v in j['other_configs'].items()}, 'limit', j['config']['records_flagged_BSX_F_LIMIT'])
"
