# GPU box: serial-mode kernel stats of one bench mode.  usage: bash tools/gpu_serial_stats.sh <tag> [bench args, e.g. --mode rrbs]
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ss_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/ss_$TAG -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 "$@" > $O/${TAG}_serial.json 2>/dev/null
cut -d, -f1-4 /tmp/ss_$TAG/s_kernel_stats.csv | cut -c1-140 | head -8 | tee $O/${TAG}_serial_stats.txt
python3 -c "
import json;d=json.load(open('$O/${TAG}_serial.json'));k=d['roofline']['dominant_kernel'];print('ms/step %.1f  hscan %.1f ms  %.0f G cand/s  heavy %d redo %d' % (d['ms_per_step'],k['ms_per_step'],k['candidates_per_s']/1e9,d['roofline']['heavy_units_last_step'],d['roofline'].get('redo_units_last_step',-1)))"
