# GPU box: SQ / TA / TCC counter summary (serial mode) of another bench mode.  usage: bash tools/profile_mode.sh <tag> --mode rrbs
# -> gpurun_out/<tag>/<tag>_sq.json (+ kernel stats); same passes as tools/profile_round.sh
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
export BSX_PROFILES_DIR=$O
# what is profiled, for the summary (bench.py quotes a summary only for the mode / counter setting it runs): --mode X and --work-counters N among the bench arguments
MODE=pe; WC=0; prev=""; for a in "$@"; do [ "$prev" = "--mode" ] && MODE=$a; [ "$prev" = "--work-counters" ] && WC=$a; prev=$a; done
export BSX_PROFILE_MODE=$MODE BSX_PROFILE_WORK_COUNTERS=$WC BSX_PROFILE_STEPS=3
S=/tmp/bsx_prof_$$; mkdir -p $S
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  t=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $S/pmc_$t -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 "$@" > /dev/null 2> $S/pmc_$t.log; echo "$t rc=$?"
done
python3 $R/tools/summarize_sq.py $TAG $S/pmc_SQ_WAVES $S/pmc_SQ_WAIT_INST_ANY $S/pmc_TA_TA_BUSY_sum > /dev/null; echo "summarize_sq rc=$?"
rm -rf $S
python3 -c "
import json
d=json.load(open('$O/${TAG}_sq.json'))
for k,v in d['kernels'].items():
    print(k, json.dumps(v['derived']))"
