# GPU box: the end-to-end command-line run on one set of files under several settings.
# usage: bash tools/gpu_e2e.sh <tag> <pairs> "<ENV=.. ENV=..[@extra options]>" ["<more settings>" ...]   ->  gpurun_out/<tag>_e2e.txt
TAG=${1:-e2e}; PAIRS=${2:-8388608}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
D=/dev/shm/bsx_e2e_$$
cd $R
python3 tools/e2e_bench.py --pairs $PAIRS --dir $D --keep > $O/${TAG}_e2e_first.json 2> $O/${TAG}_e2e_first.err; echo "first rc=$?"
python3 -c "import json;d=json.load(open('$O/${TAG}_e2e_first.json'));print('default', d['reads_per_s_mapping_phase'], d['timing']['stage_busy_s'])" | tee $O/${TAG}_e2e.txt
for s in "$@"; do
  rm -f $D/out.sam
  X=""; case "$s" in *@*) X="${s#*@}"; s="${s%%@*}";; esac   # "ENV=1 ENV=2@-p 8": settings, then extra command-line options
  env $s BSX_TIMING=1 bsmap_amd/bsmap -a $D/r_1.fq -b $D/r_2.fq -d $D/genome.fa -o $D/out.sam -s 16 -v 6 -m 28 -x 500 -S 1 $X 2> $O/${TAG}_e2e_run.err > /dev/null
  python3 - "$s" $PAIRS $O/${TAG}_e2e_run.err <<'PY' | tee -a $O/${TAG}_e2e.txt
import json, sys
t = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
print(sys.argv[1], round(2 * int(sys.argv[2]) / t["mapping_s"]), t["mapping_s"], t["stage_busy_s"], t.get("workers"))
PY
done
rm -rf $D
