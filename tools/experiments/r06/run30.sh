# round 6, call 30: k_hscan_same with a REAL register prefetch (HG_PREFETCH=1 with the first step's gathers waited for before the loop: round 5's form of the switch still waited for the
# coming step's gathers before every evaluation) — two chunks per step at four and five waves per SIMD — against the shipped form (four chunks, gathers at the top of the step)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0))"; }
run() { tag=$1; lib=$2; shift 2; BSX_LIB=$lib timeout 600 python3 bench.py --mode pe --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 6 --warmup 2 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
run base_1 bsmap_amd/libbsx.so
run pf24_1 bsmap_amd/libbsx_pf24.so
run pf25_1 bsmap_amd/libbsx_pf25.so
BSX_WORK_COUNTERS=0 BSX_LIB=bsmap_amd/libbsx_pf24.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heavy_pipeline_large_buckets or heavy_pipeline_caps" -p no:cacheprovider 2>&1 | tail -n 2
