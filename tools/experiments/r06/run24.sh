# round 6, call 24: a soak of the option fuzz (200 more draws per fuzz test, both counter routes) and whole C5 batches of 2^19 pairs against the oracle (plain and exact mode), final build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06y; mkdir -p $O; cd $R
BSX_EXTRA_FUZZ=200 timeout 2400 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "random_option or rrbs_random_options" -p no:cacheprovider > $O/fuzz.txt 2>&1; tail -n 3 $O/fuzz.txt
timeout 900 python3 tools/validate_fullsize.py --mode trim --units 524288 > $O/r06z_validate_full_c5.json 2> $O/v5.err; tail -c 600 $O/r06z_validate_full_c5.json
timeout 900 python3 tools/validate_fullsize.py --mode trim --units 524288 --exact > $O/r06z_validate_full_c5_exact.json 2> $O/v5x.err; tail -c 600 $O/r06z_validate_full_c5_exact.json
