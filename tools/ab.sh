#!/bin/bash
# A/B of kernel-tuning builds on the GPU box: one bench line per (library, workload).
# usage: bash tools/ab.sh "C2 1000 3|C1 10 20|C3 100 2" lib1.so lib2.so ...   (workload spp steps; libraries from gpurun_variants/ or the product one)
WLS=$1; shift
for lib in "$@"; do
  IFS='|' read -ra arr <<< "$WLS"
  for wl in "${arr[@]}"; do
    set -- $wl
    RTIOW_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --workload $1 --spp $2 --steps $3 --warmup 1 --no-cpu --no-sub > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$lib $wl FAILED"; tail -3 gpurun_out/ab_tmp.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('%-28s %-3s %5s spp: %8.0f Mrays/s %9.2f ms' % ('$(basename $lib)', '$1', '$2', d['value'], d['ms_per_step']))"
  done
done
