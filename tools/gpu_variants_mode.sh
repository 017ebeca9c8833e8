# GPU box: default-mode bench line of one bench mode for the default library and named variants.  usage: bash tools/gpu_variants_mode.sh <tag> "<bench args>" [variant ...]
TAG=$1; ARGS=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in default "$@"; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  timeout 600 python3 bench.py $ARGS --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/${TAG}_$v.json 2> $O/${TAG}_$v.err
  python3 -c "import json;d=json.load(open('$O/${TAG}_$v.json'));print('$v: ms/step %.1f  reads/s %.0f' % (d['ms_per_step'], d['value']))"
done
