# GPU box: the driver threads after the change from hipEventSynchronize to poll + sleep.  usage: bash tools/experiments/r03/nospin.sh <tag>
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
python3 tools/driver_cpu.py > $O/driver_cpu.json 2> $O/driver_cpu.err; cut -c1-500 $O/driver_cpu.json
for nf in 2 1; do
python3 bench.py --cpu-seconds 0 --e2e-pairs 0 --sensitivity 0 --other-configs 0 --in-flight $nf > $O/bench_f$nf.json 2> $O/bench_f$nf.err
python3 -c "
import json; d=json.load(open('$O/bench_f$nf.json')); tr=d.get('value_incl_transfers') or {}
print('in flight $nf: %.1f ms/step %.2f M; incl transfers %s ms/step' % (d['ms_per_step'], d['value']/1e6, tr.get('ms_per_step')))"
done
for m in trim rrbs; do
python3 bench.py --mode $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > $O/bench_$m.json 2> $O/bench_$m.err
python3 -c "
import json; d=json.load(open('$O/bench_$m.json')); print('$m: %.1f ms/step %.2f M' % (d['ms_per_step'], d['value']/1e6))"
done
bash tools/host_ceiling.sh $TAG 2>&1 | tail -3
python3 -c "
import json
for g in ('0.002','1.0'):
    t=json.load(open('gpurun_out/${TAG}_hc_%s.json' % g))['timing']; print(g, t['mapping_s'], t['mapping_cpu_s'])"
