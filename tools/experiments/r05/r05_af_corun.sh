# GPU box: now that both big kernels are bound by vector issue (0.66 / 0.71 busy alone), does leaving room for the other batch's scan kernel beside the main kernel pay?
# k_align at N waves per CU (default 20 = five per SIMD), two and three batches in flight
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05af; mkdir -p $O; cd $R
for w in 0 16 12 8; do for f in 2 3; do
  timeout 600 python3 bench.py --in-flight $f $([ $w != 0 ] && echo --waves-per-cu $w) --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 12 --warmup 3 > $O/w${w}_f$f.json 2>/dev/null
  python3 -c "
import json
try:
    d=json.load(open('$O/w${w}_f$f.json')); print('waves/CU $w, in flight $f: %.1f ms/step  %.2f M reads/s' % (d['ms_per_step'], d['value']/1e6))
except Exception as e: print('waves/CU $w in flight $f failed', e)"
done; done
