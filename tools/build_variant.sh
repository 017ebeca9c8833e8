#!/bin/bash
# build a tuning variant of libbsx.so: tools/build_variant.sh <name> <extra -D flags...>  ->  bsmap_amd/libbsx_<name>.so
# (run the bench against it with BSX_LIB=bsmap_amd/libbsx_<name>.so; bsx_align.hip and bsx_api.hip are recompiled)
set -e
N=$1; shift
cd "$(dirname "$0")/../bsmap_amd/csrc"
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c bsx_align.hip -o /tmp/bsx_align_$N.o &
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result "$@" -c bsx_api.hip -o /tmp/bsx_api_$N.o   # (shares bsx_kernel_args.h with the kernels)
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libbsx_$N.so bsx_host.o /tmp/bsx_api_$N.o /tmp/bsx_align_$N.o bsx_index.o bsx_synth.o bsx_meth.o bsx_probe.o bsx_pack.o -lz -Wl,-rpath,/opt/rocm/lib
echo built bsmap_amd/libbsx_$N.so
