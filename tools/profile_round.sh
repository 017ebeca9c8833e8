# GPU box: the round's artefacts for the CURRENT build.  usage: bash tools/profile_round.sh <tag>   (e.g. r02z)
#   <tag>_bench*.json          default bench line (C3, all legs) and the C2 / C4 / C5 lines
#   <tag>_stats/               rocprofv3 --kernel-trace --stats of the serial-mode bench (no two kernels overlap)
#   <tag>_stats_default/       the same of the default (two batches in flight) bench
#   <tag>_pmc_*/               SQ / TA+TCC / FETCH_SIZE / WRITE_SIZE counter passes (serial mode, separate runs)
# then, back in the container:  python3 tools/summarize_pmc.py <tag> gpurun_out/<tag>_stats gpurun_out/<tag>_pmc_FETCH_SIZE gpurun_out/<tag>_pmc_WRITE_SIZE
#                               python3 tools/summarize_sq.py <tag> gpurun_out/<tag>_pmc_SQ_WAVES gpurun_out/<tag>_pmc_SQ_WAIT_INST_ANY gpurun_out/<tag>_pmc_TA_TA_BUSY_sum
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
bash $R/tools/bench_all.sh $TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_bench_serial.json 2> $O/${TAG}_stats.log; echo "stats rc=$?"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats_default -o s --output-format csv -- python3 $R/bench.py --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 --steps 4 --warmup 2 > $O/${TAG}_bench_under_rocprof.json 2> $O/${TAG}_stats_default.log; echo "stats default rc=$?"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $O/${TAG}_pmc_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 > $O/${TAG}_pmc_$t.json 2> $O/${TAG}_pmc_$t.log; echo "$t rc=$?"
done
$R/tools/microbench/valu_issue > $O/${TAG}_valu_issue.json 2>/dev/null; echo "valu rc=$?"
