#!/bin/bash
# usage: tools/build_variants.sh name1 "-DA=1 -DB=2" name2 "-D..." ...   -> gpurun_variants/lib_NAME.so (parallel builds)
cd "$(dirname "$0")/.."; mkdir -p gpurun_variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Iinclude"
while [ $# -ge 2 ]; do
  ( /opt/rocm/bin/hipcc $FLAGS $2 -c -o gpurun_variants/obj_$1.o raytracing-in-one-weekend_amd/csrc/rt_device.hip 2>&1 | grep -E "error"; /opt/rocm/bin/hipcc -fPIC --offload-arch=gfx950 -shared -o gpurun_variants/lib_$1.so gpurun_variants/obj_$1.o raytracing-in-one-weekend_amd/csrc/.obj/rt_bvh_build.o raytracing-in-one-weekend_amd/csrc/.obj/rt_collective.o -ldl; rm -f gpurun_variants/obj_$1.o ) &
  shift 2
done
wait; ls gpurun_variants
