#!/bin/bash
# A/B builds of render_fused.hip with experiment macros: scripts/build_variant.sh <name> "<-D flags>" -> nerfsafetyvalidation_amd/libngp_hip_<name>.so
# (select at run time with NGP_HIP_LIB=$PWD/nerfsafetyvalidation_amd/libngp_hip_<name>.so)
set -e
cd "$(dirname "$0")/../nerfsafetyvalidation_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function -DNGP_BUILD $@ -c render_fused.hip -o /tmp/render_fused_$name.o
objs=$(ls *.o | grep -v render_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libngp_hip_$name.so $objs /tmp/render_fused_$name.o
echo built ../libngp_hip_$name.so
