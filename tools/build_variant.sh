#!/bin/bash
# usage: tools/build_variant.sh <name> <kernel-file-stem> "<extra -D flags>"  -> build/variants/libtb_<name>.so
# Experiment builds: one kernel file recompiled with extra flags, linked with the tree's other objects; select with TB_HIP_LIB.
set -e
name=$1; stem=$2; extra=$3
root=$(cd "$(dirname "$0")/.." && pwd)
cs=$root/trackingbench_slam_amd/csrc
mkdir -p $root/build/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-value $extra -c $cs/$stem.hip -o $root/build/variants/${stem}_$name.o
# the host file shares tb_internal.h with the kernels (block geometry macros such as FB_TH size both the host's block table and
# the kernel's LDS tile): it is rebuilt with the same flags, so a variant can never pair a kernel with a table of another geometry
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -Wno-unused-value $extra -x hip -c $cs/tb_capi.cpp -o $root/build/variants/tb_capi_$name.o
objs=""
for o in k_pyramid k_fast k_octree k_describe k_match k_pose k_ba k_flow k_ransac; do
  if [ "$o" = "$stem" ]; then objs="$objs $root/build/variants/${stem}_$name.o"; else objs="$objs $cs/$o.o"; fi
done
objs="$objs $root/build/variants/tb_capi_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $root/build/variants/libtb_$name.so $objs
echo $root/build/variants/libtb_$name.so
