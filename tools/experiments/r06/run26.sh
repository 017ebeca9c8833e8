# round 6, calls 26 and 31: the whole -m gpu suite and the round's profile set for the build with shared_window<3> (26), and again with its split fetch and two chunks per step (31)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu -p no:cacheprovider > gpurun_out/r06z_gpu_suite.txt 2>&1; tail -n 3 gpurun_out/r06z_gpu_suite.txt
bash tools/profile_round.sh r06z > gpurun_out/r06z_round.log 2>&1; tail -n 14 gpurun_out/r06z_round.log
timeout 600 python3 tools/validate_fullsize.py --mode trim --units 524288 > gpurun_out/r06z/r06z_validate_full_c5.json 2>/dev/null; echo "validate c5 rc=$?"
