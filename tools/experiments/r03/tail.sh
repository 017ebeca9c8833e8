# GPU box: tail mode of the heavy pipeline (small scan grids on the control stream) — parity subset, bench lines, command-line timeline.  usage: bash tools/experiments/r03/tail.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_synth.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for tt in 16384 0; do
  for nf in 2 1; do
    BSX_TAIL_TASKS=$tt python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 --in-flight $nf > $O/bench_t${tt}_f$nf.json 2> $O/bench_t${tt}_f$nf.err
    python3 -c "
import json; d=json.load(open('$O/bench_t${tt}_f$nf.json')); tr=d.get('value_incl_transfers') or {}
print('tail_tasks $tt in flight $nf: %.1f ms/step %.2f M; incl transfers %s ms/step' % (d['ms_per_step'], d['value']/1e6, tr.get('ms_per_step')))"
  done
  BSX_TAIL_TASKS=$tt BSX_TIMING=2 python3 tools/e2e_bench.py --pairs 16777216 --genome 1.0 --dir /dev/shm/bsx_t_$$ > $O/e2e_t$tt.json 2> $O/e2e_t$tt.err
  python3 -c "
import json
d=json.load(open('$O/e2e_t$tt.json')); t=d['timing']; n=2*d['pairs']
print('tail_tasks $tt e2e: %.2f M reads/s  mapping %.2f s' % (n/t['mapping_s']/1e6, t['mapping_s']), {k: t['stage_busy_s'][k] for k in ('gpu','gpu_align','format','write')})"
done
python3 tools/e2e_gantt.py $O/e2e_t16384.json | head -19
