# usage: bash tools/trace_step.sh <tag> <groups> <in-flight> <mode> [n] [skip]
TAG=$1; G=$2; NF=$3; M=$4; N=${5:-120}; SK=${6:-0}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=/tmp/tr_$TAG; rm -rf $S
BSX_HEAVY_GROUPS=$G rocprofv3 --kernel-trace -d $S -o t --output-format csv -- python3 $R/bench.py --mode $M --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 3 --warmup 1 --in-flight $NF > $O/bench.json 2> $O/bench.err; echo rc=$?
python3 $R/tools/timeline.py $S > $O/timeline.json; cut -c1-900 $O/timeline.json
python3 $R/tools/timeline_dump.py $S $N $SK > $O/dump.txt; head -150 $O/dump.txt
