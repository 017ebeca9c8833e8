# GPU box: vector instructions of the main kernel by phase: SQ_INSTS_VALU of builds that stop a unit behind unit_prepare (prep1) / in front of unit_finish (prep2)
# (-DBSX_EXP_PREPONLY=1 / 2; their results are garbage, only the instruction counts are read) against the whole kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ae; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in ${VARS:-default prep1 prep2}; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  rm -rf /tmp/vp_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU --kernel-trace -d /tmp/vp_$v -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> /tmp/vp_$v.log; echo "$v rc=$?"
  python3 - <<PY
import csv,collections
tot=collections.defaultdict(float); n=0
for row in csv.DictReader(open('/tmp/vp_$v/p_counter_collection.csv')):
    if 'k_align' in row['Kernel_Name']: tot[row['Counter_Name']]+=float(row['Counter_Value'])
units=3*(1<<22)
print('$v per unit:', {k: round(x/units,1) for k,x in tot.items()})
PY
done 2>&1 | tee $O/valu_by_phase.txt
