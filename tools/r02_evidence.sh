# round-2 evidence pass (GPU box): VALU issue microbenchmark, memory ceilings, serial-mode kernel stats and SQ/TA passes.
# usage: bash tools/r02_evidence.sh <tag>      -> gpurun_out/<tag>_*
TAG=${1:-r02a}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out
mkdir -p $O
$R/tools/microbench/valu_issue > $O/${TAG}_valu_issue.json 2> $O/${TAG}_valu_issue.err; echo "valu rc=$?"
python3 -c "
import sys, json; sys.path.insert(0, '$R')
import torch, bsmap_amd as B
print(json.dumps(B.probe_memory(0, 4 << 30, 1 << 30)))" > $O/${TAG}_probe_memory.json 2> $O/${TAG}_probe.err; echo "probe rc=$?"
rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 > $O/${TAG}_bench_serial.json 2> $O/${TAG}_stats.log; echo "stats rc=$?"
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $O/${TAG}_pmc_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 1 --warmup 1 > $O/${TAG}_pmc_$t.json 2> $O/${TAG}_pmc_$t.log; echo "$t rc=$?"
done
