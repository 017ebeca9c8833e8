# GPU box: per-pass table of the heavy pipeline in serial mode.  usage: bash tools/pass_profile.sh <tag> <mode> [bench args]  ->  gpurun_out/<tag>/passes_<mode>.txt
TAG=$1; M=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=/tmp/pp_${TAG}_$M; rm -rf $S
BSX_TRACE_HEAVY=1 rocprofv3 --kernel-trace -d $S -o t --output-format csv -- python3 $R/bench.py --mode $M --profile-serial --steps 2 --warmup 1 "$@" > $O/serial_$M.json 2> $O/serial_$M.err; echo "$M rc=$?"
python3 $R/tools/pass_table.py $S $O/serial_$M.err > $O/passes_$M.txt; head -1 $O/passes_$M.txt
rm -rf $S
