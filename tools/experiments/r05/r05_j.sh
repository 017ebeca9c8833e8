# GPU box, round 5 (j): can k_align (fabric-bound, vector units idle) and the scan kernel (issue-bound, now asking the fabric for less) share the CUs?
# k_align limited to W waves per CU (bsx_set_waves_per_cu) so that scan waves of the other batches in flight fit beside it; segments on / off
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O; cd $R
run() { # name, args...
  n=$1; shift
  timeout 900 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 "$@" > $O/$n.json 2> $O/$n.err
  python3 -c "
import json
try:
    d=json.load(open('$O/$n.json')); k=d['roofline']['dominant_kernel']
    print('$n: %.1f ms/step  %.2f M reads/s   scan %.1f ms/step serial %.1f' % (d['ms_per_step'], d['value']/1e6, k['ms_per_step'], d['roofline']['serial_replay']['ms_per_step']))
except Exception as e: print('$n failed', e); print(open('$O/$n.err').read()[-400:])"
}
for seg in 1 0; do
  export BSX_SEG=$seg
  run seg${seg}_w20_f2 --steps 4 --warmup 2
  run seg${seg}_w12_f2 --steps 4 --warmup 2 --waves-per-cu 12
  run seg${seg}_w8_f2 --steps 4 --warmup 2 --waves-per-cu 8
  run seg${seg}_w8_f3 --steps 6 --warmup 3 --waves-per-cu 8 --in-flight 3
  run seg${seg}_w12_f3 --steps 6 --warmup 3 --waves-per-cu 12 --in-flight 3
  run seg${seg}_w4_f3 --steps 6 --warmup 3 --waves-per-cu 4 --in-flight 3
done
