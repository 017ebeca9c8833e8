# GPU box: serial-mode kernel stats + SQ / TA / TCC / FETCH / WRITE counter passes of the current build.  usage: bash tools/r02_pmc.sh <tag>
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_bench_serial.json 2> $O/${TAG}_stats.log; echo "stats rc=$?"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $O/${TAG}_pmc_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 > $O/${TAG}_pmc_$t.json 2> $O/${TAG}_pmc_$t.log; echo "$t rc=$?"
done
