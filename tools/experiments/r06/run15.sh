# round 6, call 15: the whole -m gpu suite on the round's build, then the driver's bench command
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06o; mkdir -p $O; cd $R
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.txt 2>&1; tail -n 5 $O/gpu_suite.txt
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; echo "bench rc=$?"; tail -n 3 $O/bench_driver_command.err
python3 - <<'PY'
import json
d=json.load(open("gpurun_out/r06o/bench_driver_command.json"))
print("value %.2f M  ms/step %.1f" % (d["value"]/1e6, d["ms_per_step"]))
r=d["roofline"]; print({k:r[k] for k in ("bound","achieved","peak","frac","issue_utilisation","instructions_per_evaluation","binding_kernel","serial_ms_k_align","serial_ms_k_hctrl","serial_ms_order","serial_ms_scan","serial_control_passes")})
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","cores","kind","port_over_reference","value_reference_equivalent")})
for k,v in d.get("other_configs",{}).items(): print(k, {a:v.get(a) for a in ("reads_per_s","ms_per_step","serial_ms","binding_kernel","control_passes","error")})
e=d.get("end_to_end",{}); print("e2e", {k:e.get(k) for k in ("reads_per_s","mapping_s","load_reference_s","whole_process_s","best_reads_per_s","error")}, (e.get("two_lanes_one_gpu") or {}).get("reads_per_s"))
print("incl transfers", (d.get("value_incl_transfers") or {}).get("value"))
PY
