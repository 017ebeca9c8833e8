# round 6, call 17: the device packer: its tests, the reference tests of the parity suite, the bench's FASTA path at size, start-up of the command line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06q; mkdir -p $O; cd $R
true
V=";BSX_HOST_PACK=1"
timeout 2400 python3 tools/e2e_bench.py --pairs 4194304 --dir /dev/shm/bsx_e2e_$$ --variants "$V" > $O/e2e.json 2> $O/e2e.err; echo rc=$?; tail -n 3 $O/e2e.err
grep device_pack_s $O/e2e.err | head -3; python3 -c "
import json
d=json.load(open('$O/e2e.json'))
for r in d.get('variants', [d]):
    t=r['timing']; n=2*r['pairs']
    print(t.get('device_pack')); print('[%-20s] load_reference %.2f s  index %.2f s  mapping %.2f s  whole %.2f s' % (r.get('cli_args',''), t.get('load_reference_s',0), t.get('index_build_s',0), t['mapping_s'], r['cli_wall_s']))"
