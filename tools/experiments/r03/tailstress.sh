# GPU box: whole-batch parity with the tail mode forced early / tiny sweeping grids / two unit groups.  usage: bash tools/experiments/r03/tailstress.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
i=0
for env in "BSX_TAIL_TASKS=100000000 BSX_TAIL_GRID=64" "BSX_TAIL_TASKS=100000000 BSX_TAIL_GRID=1000" "BSX_HEAVY_GROUPS=2" "BSX_HEAVY_GROUPS=3 BSX_TAIL_TASKS=100000000"; do
  i=$((i+1))
  env $env timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_leak_exact.py -m gpu -x -q > $O/pytest_$i.log 2>&1; echo "$env: rc=$? $(tail -1 $O/pytest_$i.log)"
done
