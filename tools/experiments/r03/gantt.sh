# GPU box: stage timeline of the command line at hg38 size.  usage: bash tools/experiments/r03/gantt.sh <tag> [device batches]
TAG=$1; NB=${2:-2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
BSX_TIMING=2 BSX_GPU_BATCHES=$NB python3 tools/e2e_bench.py --pairs 16777216 --genome 1.0 --dir /dev/shm/bsx_g_$$ > $O/e2e_nb$NB.json 2> $O/e2e_nb$NB.err
python3 tools/e2e_gantt.py $O/e2e_nb$NB.json
