# GPU box: instruction-cache counters of the main kernel (19.5 K instructions: four inlined copies of the list scan), serial mode
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05ad; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
[ -n "$LIBV" ] && export BSX_LIB=$R/bsmap_amd/libbsx_$LIBV.so
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INST_LEVEL_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA"; do
  t=$(echo $set | cut -d' ' -f1)
  rm -rf /tmp/ic_$t
  rocprofv3 --pmc $set --kernel-trace -d /tmp/ic_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> /tmp/ic_$t.log; echo "$t rc=$?"
  python3 - <<PY
import csv,collections
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for row in csv.DictReader(open('/tmp/ic_$t/p_counter_collection.csv')):
    k=row['Kernel_Name']
    for name in ('k_align','k_hscan_same','k_hctrl'):
        if name in k: tot[name][row['Counter_Name']]+=float(row['Counter_Value'])
for name in tot: print(name, dict(tot[name]))
PY
done 2>&1 | tee $O/icache_${LIBV:-default}.txt
