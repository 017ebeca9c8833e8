# round 6, call 5: get_pairs with the b hits staged in LDS: parity (PE heavy cases, golden, fuzz without counters), clocks, lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06e; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "heavy or golden or context" > $O/parity.txt 2>&1; tail -n 3 $O/parity.txt
for m in trim; do timeout 600 python3 tools/ctrl_clocks.py --mode $m > $O/ctrl_clocks_${m}.json 2> $O/ctrl_clocks_${m}.err; cut -c1-1200 $O/ctrl_clocks_${m}.json; echo; done
for m in trim; do bash tools/pass_profile.sh r06e_new $m --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0; done
bash tools/ab_libs.sh r06e_ab r05 "trim rrbs" 1
