# round 6, call 25: k_hscan_shared with the 65-96 nt form of its window loop (shared_window<3>: two row units, plain reads two v_bitop3 per inner word, no absent words) against
# -DBSX_HSHARED_W3=0, and at six waves per SIMD; RRBS parity first
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_synth.py -x -q -m gpu -k "rrbs" -p no:cacheprovider > $O/w3_tests.txt 2>&1; tail -n 3 $O/w3_tests.txt
line() { python3 -c "
import json
d=json.load(open('$1')); r=d['roofline']; print('$2: %.1f ms/step  %.2f M reads/s   serial %.1f  align %.1f ctrl %.1f order %.1f scan %.1f passes %.0f' % (d['ms_per_step'], d['value']/1e6, r['serial_ms_per_step'] or 0, r['serial_ms_k_align'] or 0, r['serial_ms_k_hctrl'] or 0, r['serial_ms_order'] or 0, r['serial_ms_scan'] or 0, r['serial_control_passes'] or 0))"; }
run() { tag=$1; lib=$2; shift 2; BSX_LIB=$lib timeout 900 python3 bench.py --mode rrbs --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 9 --warmup 3 "$@" > $O/$tag.json 2> $O/$tag.err; line $O/$tag.json "$tag" || tail -n 3 $O/$tag.err; }
for rep in 1 2; do
  run w3_$rep bsmap_amd/libbsx.so
  run now3_$rep bsmap_amd/libbsx_now3.so
  run w3x6_$rep bsmap_amd/libbsx_w3x6.so
done
