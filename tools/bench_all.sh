# GPU box: the default bench line (C3, all legs) and the C2 / C4 / C5 lines.  usage: bash tools/bench_all.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout 1200 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err; echo "default rc=$?"
for m in se rrbs trim; do
  timeout 900 python3 bench.py --mode $m --e2e-pairs 0 > $O/${TAG}_bench_$m.json 2> $O/${TAG}_bench_$m.err; echo "$m rc=$?"
done
python3 - <<PY
import json
for f in ("${TAG}_bench", "${TAG}_bench_se", "${TAG}_bench_rrbs", "${TAG}_bench_trim"):
    try:
        d = json.load(open("$O/" + f + ".json"))
        cb = d.get("cpu_baseline") or {}
        print(f, "%.2f M reads/s  %.1f ms/step  incl-transfers %s  cpu %s p8 %s  e2e %s" % (d["value"] / 1e6, d["ms_per_step"],
              (d.get("value_incl_transfers") or {}).get("value"), cb.get("value"), (cb.get("p8") or {}).get("value"), (d.get("end_to_end") or {}).get("reads_per_s")))
    except Exception as e:
        print(f, "no line:", e)
PY
