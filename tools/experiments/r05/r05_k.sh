# GPU box, round 5 (k): fabric read requests and L2 hit of the group scan with and without segments (one TCC counter pass each, serial mode)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
export BSX_PROFILES_DIR=$O BSX_PROFILE_MODE=pe BSX_PROFILE_WORK_COUNTERS=0 BSX_PROFILE_STEPS=3 BSX_PROFILE_UNITS=4194304
cd /tmp && export TMPDIR=/tmp
for seg in 1 0; do
  export BSX_SEG=$seg
  S=/tmp/bsx_tcc_$seg; rm -rf $S
  rocprofv3 --pmc TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace -d $S -o p --output-format csv -- python3 $R/bench.py --profile-serial --steps 2 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> $O/tcc_$seg.log; echo "seg $seg rc=$?"
  python3 $R/tools/summarize_sq.py seg$seg $S > /dev/null
  python3 -c "
import json
j=json.load(open('$O/seg${seg}_sq.json'))
for k,v in j['kernels'].items():
    d=v['derived']; print('seg=$seg', k, 'fabric requests per step %.3g' % d.get('fabric_read_requests_per_step',0), 'L2 hit %.3f' % d.get('l2_hit_frac',0), 'TA busy %.3f' % d.get('ta_busy_frac',0), 'L1 miss/access %.3f' % d.get('l1_miss_per_access',0))"
done
