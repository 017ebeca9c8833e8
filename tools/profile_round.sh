cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/bench.py > $R/gpurun_out/bench_default.json 2> $R/gpurun_out/bench_default.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f_stats -o s --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --e2e-pairs 0 > $R/gpurun_out/r01f_stats.log 2>&1; echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/r01f_fetch -o f --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --e2e-pairs 0 --steps 2 --warmup 1 --in-flight 1 > $R/gpurun_out/r01f_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/r01f_write -o w --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --e2e-pairs 0 --steps 2 --warmup 1 --in-flight 1 > $R/gpurun_out/r01f_write.log 2>&1; echo "write rc=$?"
