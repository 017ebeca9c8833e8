# GPU box, round 5 (a): what the class statistics of k_hscan_same cost (A/B against a build without them), and how many evaluations repeat an
# earlier member's read words inside their group (BSX_SIGHIST)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O; cd $R
bash tools/ab_libs.sh r05b nostats "pe se" 2
BSX_SIGHIST=1 timeout 600 python3 bench.py --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 2 --warmup 1 2>&1 >/dev/null | grep sighist > $O/sighist_pe.txt; grep -v "R <=" $O/sighist_pe.txt
BSX_SIGHIST=1 timeout 600 python3 bench.py --mode se --in-flight 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --sensitivity 0 --other-configs 0 --steps 2 --warmup 1 2>&1 >/dev/null | grep sighist > $O/sighist_se.txt; grep -v "R <=" $O/sighist_se.txt
