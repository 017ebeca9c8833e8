cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmcb_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 > $R/gpurun_out/pmcb_$tag.log 2>&1
  echo "$tag rc=$?"
done
