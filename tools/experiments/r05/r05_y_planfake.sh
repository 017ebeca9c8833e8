# GPU box: what the planner's bucket-size lookups cost the main kernel.  libbsx_planfake.so (tools/build_variant.sh planfake -DBSX_EXP_PLANFAKE, a patch of
# plan_counts that reads bucket_off[key & 0xffff]: wrong plans, same amount of work) against the shipped library, serial mode, per-kernel times from rocprofv3.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in default planfake planfake23 default planfake planfake23; do
  if [ $v = default ]; then unset BSX_LIB; else export BSX_LIB=$R/bsmap_amd/libbsx_$v.so; fi
  rm -rf /tmp/pf_$v
  rocprofv3 --kernel-trace --stats -d /tmp/pf_$v -o s --output-format csv -- python3 $R/bench.py --profile-serial --steps 3 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 > /dev/null 2> /tmp/pf_$v.log
  echo "== $v"; grep -E "k_align|k_hscan_same|k_hctrl" /tmp/pf_$v/s_kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
done | tee $O/planfake.txt
