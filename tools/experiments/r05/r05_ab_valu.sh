# GPU box: the main kernel's VALU diet (wave_min_u32 by DPP, start totals over four lane groups, the counted LOCATE) against the build before it (libbsx_old.so)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ab; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for rep in 1 2; do for v in default old $EXTRA; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  for m in pe se; do
    timeout 600 python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 12 --warmup 3 > $O/${m}_${v}_$rep.json 2>/dev/null
    python3 -c "
import json
d=json.load(open('$O/${m}_${v}_$rep.json')); print('$m $v #$rep: %.1f ms/step  %.2f M reads/s  serial %.1f' % (d['ms_per_step'], d['value']/1e6, d['roofline']['serial_replay']['ms_per_step']))"
  done
done; done
cd /tmp && export TMPDIR=/tmp
for v in default old $EXTRA; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  rm -rf /tmp/pf_$v
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> /tmp/pf_$v.log
  echo "== $v"; grep -E "k_align|k_hscan_same|k_hctrl" /tmp/pf_$v/s_kernel_stats.csv | sed 's/"[^"]*k_\([a-z_]*\)[^"]*"/\1/' | cut -d, -f1-4
done
