# GPU box: where a wave of the main kernel spends its clocks (libbsx_phase.so = align_phase_clocks.patch built with tools/build_variant.sh phase -DBSX_EXP_PHASE:
# s_memtime around the parts of process_unit / snp_align / wave_scan_range, summed per wave, printed when the batch closes)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
export BSX_LIB=$R/bsmap_amd/libbsx_phase.so BSX_SECTOR_STATS=1
for m in pe se; do
  python3 bench.py --mode $m --profile-serial --steps 3 --warmup 1 --cpu-seconds 0 --e2e-pairs 0 --transfer-steps 0 2>&1 >/dev/null | grep "\[phase\]" > $O/phase_$m.txt
  echo "== $m"; cat $O/phase_$m.txt
done
