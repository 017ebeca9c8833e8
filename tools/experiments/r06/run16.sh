# round 6, call 16: the FASTA packer alone on the box's host CPUs at hg38 size: round 5's (one thread per record) against the chunk-parallel one
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06p; mkdir -p $O; cd $R
python3 - <<'PY'
import numpy as np
HG38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]
rng = np.random.default_rng(1)
with open("/dev/shm/pack.fa", "wb") as f:
    for c, n in enumerate(HG38):
        f.write(b">chr%d\n" % (c + 1))
        a = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]
        a[n // 3:n // 3 + n // 20] = ord("N")
        full = n // 60 * 60
        rows = np.empty((full // 60, 61), np.uint8); rows[:, :60] = a[:full].reshape(-1, 60); rows[:, 60] = 10
        rows.tofile(f)
PY
ls -la /dev/shm/pack.fa
cp tools/experiments/r06/pack_time/bsx_host_r05.cpp.txt /tmp/bsx_host_r05.cpp
sed -i "s|/root/repo/bsmap_amd/csrc|$R/bsmap_amd/csrc|g" /tmp/bsx_host_r05.cpp
/opt/rocm/bin/hipcc -O3 -std=c++17 -x hip --cuda-host-only -pthread -DPACK_SRC="\"$R/bsmap_amd/csrc/bsx_host.cpp\"" -o /tmp/pack_new tools/experiments/r06/pack_time/pack_time.cpp 2>&1 | grep error
/opt/rocm/bin/hipcc -O3 -std=c++17 -x hip --cuda-host-only -pthread -DPACK_SRC="\"/tmp/bsx_host_r05.cpp\"" -o /tmp/pack_old tools/experiments/r06/pack_time/pack_time.cpp 2>&1 | grep error
NODE_CPUS=$(cat /sys/devices/system/node/node0/cpulist | cut -d, -f1 | cut -d- -f1)
for v in old new old new; do echo "== $v (16 CPUs of node 0)"; taskset -c 0-15 /tmp/pack_$v /dev/shm/pack.fa; done 2>&1 | tee $O/pack_time.txt
for v in old new; do echo "== $v (unpinned)"; /tmp/pack_$v /dev/shm/pack.fa; done 2>&1 | tee -a $O/pack_time.txt
rm -f /dev/shm/pack.fa
