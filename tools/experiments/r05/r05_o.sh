# GPU box, round 5 (o): batches in flight / unit groups at 2^22 pairs per step with the final scan kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O; cd $R
run() { n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']
    print('$n: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f pools %s' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], (d['roofline'].get('serial_replay') or {}).get('ms_per_step') or 0, d['config']['heavy_pools'][0]))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-300:])"
}
run f2 --steps 8 --warmup 2
run f3 --steps 9 --warmup 3 --in-flight 3
BSX_HEAVY_GROUPS=2 run f2_g2 --steps 8 --warmup 2
run f2_again --steps 8 --warmup 2
run f4_3m --steps 8 --warmup 4 --in-flight 4 --pairs-per-step 3145728
run f2_6m --steps 6 --warmup 2 --pairs-per-step 6291456
