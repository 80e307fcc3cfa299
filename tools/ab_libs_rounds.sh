#!/bin/bash
# A/B of library builds in alternating rounds.  usage: bash tools/ab_libs_rounds.sh ROUNDS "C1 10 30|C3 100 2" libA.so libB.so ...
# ("product" = raytracing-in-one-weekend_amd/lib/librtiow_hip.so)
ROUNDS=$1; WLS=$2; shift 2
LIBS=("$@")
for ((r = 0; r < ROUNDS; ++r)); do
  for lib in "${LIBS[@]}"; do
    path=$PWD/$lib; [ "$lib" = product ] && path=$PWD/raytracing-in-one-weekend_amd/lib/librtiow_hip.so
    IFS='|' read -ra arr <<< "$WLS"
    for wl in "${arr[@]}"; do
      set -- $wl
      RTIOW_HIP_LIB=$path timeout -k 10 400 python3 bench.py --workload $1 --spp $2 --steps $3 --warmup 1 --no-cpu --no-sub > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$lib $wl FAILED"; tail -3 gpurun_out/ab_tmp.err; exit 1; }
      python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('round $r  %-28s %-3s %5s spp: %8.0f Mrays/s %9.3f ms' % ('$(basename $lib)', '$1', '$2', d['value'], d['ms_per_step']))"
    done
  done
done
