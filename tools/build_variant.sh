#!/bin/bash
# A variant of the library for A/B runs on one GPU box (tools/ab_perf.sh, DSB_LIB_PATH): tools/build_variant.sh NAME [hipcc flags...]
# -> ab/NAME.so (ab/ is not tracked; it travels with gpurun).  E.g.  tools/build_variant.sh w4 -DDSB_WAVES_PER_EU=4 -DDSB_LDS_DIET;  -DDSB_NO_INLINE: the quick build (dsb_wave.h)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p ab/obj_$name
F="-O3 -fno-strict-aliasing --offload-arch=gfx950 -fPIC -std=c++17 -Wno-unused-value -Iinclude $*"
rm -f ab/obj_$name/*.o
for u in dsb_index.cpp dsb_build.hip; do /opt/rocm/bin/hipcc $F -c desamba_amd/csrc/$u -o ab/obj_$name/$u.o & done
for k in 0 1 2 3 4; do /opt/rocm/bin/hipcc $F -DDSB_KUNIT=$k -c desamba_amd/csrc/dsb_gpu.hip -o ab/obj_$name/dsb_gpu.hip.k$k.o & done     # (five units side by side: dsb_gpu.hip)
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/$name.so ab/obj_$name/*.o -lz
echo "built ab/$name.so"
