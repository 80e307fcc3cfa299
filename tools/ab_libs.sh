#!/bin/bash
# A/B of kernel-tuning builds: bench lines of the given workloads with each library in raytracing-in-one-weekend_amd/lib/exp_*.so
# next to the product library (RTIOW_HIP_LIB picks the library).  Usage on the box: bash tools/ab_libs.sh > gpurun_out/ab.txt
set -e
L=raytracing-in-one-weekend_amd/lib
for lib in $L/librtiow_hip.so $L/exp_*.so; do
  for wl in "C2 1000" "C1 100" "C3 128"; do
    set -- $wl
    echo -n "$(basename $lib) $1 $2spp: "
    RTIOW_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --workload $1 --spp $2 --no-cpu --steps 3 --warmup 1 | python -c 'import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print("%.1f Mrays/s  %.2f ms" % (d["value"], d["ms_per_step"]))'
  done
done
