# GPU box: PE only, variants named in $VARS against libbsx_old.so: two batches in flight and the serial per-kernel times
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ac; mkdir -p $O; cd $R
for rep in 1 2; do for v in old $VARS; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 12 --warmup 3 > $O/pe_${v}_$rep.json 2>/dev/null
  python3 -c "
import json
d=json.load(open('$O/pe_${v}_$rep.json')); print('pe $v #$rep: %.1f ms/step  %.2f M reads/s  serial %.1f' % (d['ms_per_step'], d['value']/1e6, d['roofline']['serial_replay']['ms_per_step']))"
done; done
cd /tmp && export TMPDIR=/tmp
for v in old $VARS; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  rm -rf /tmp/pf_$v
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> /tmp/pf_$v.log
  echo "== $v"; grep -E "k_align" /tmp/pf_$v/s_kernel_stats.csv | sed 's/"[^"]*k_\([a-z_]*\)[^"]*"/\1/' | cut -d, -f1-4
done
