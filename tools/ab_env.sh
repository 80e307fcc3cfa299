#!/bin/bash
# A/B of an environment switch on the GPU box: one bench line per (setting, workload).
# usage: bash tools/ab_env.sh "C1 10 20|C1 100 5" "RT_WF_LDS_TREE=0" "RT_WF_LDS_TREE=1" ...   (workload spp steps)
WLS=$1; shift
for setting in "$@"; do
  IFS='|' read -ra arr <<< "$WLS"
  for wl in "${arr[@]}"; do
    set -- $wl
    env RT_KNOBS=1 $setting timeout -k 10 300 python3 bench.py --workload $1 --spp $2 --steps $3 --warmup 1 --no-cpu --no-sub > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$setting $wl FAILED"; tail -3 gpurun_out/ab_tmp.err; exit 1; }
    python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('%-28s %-3s %5s spp: %8.0f Mrays/s %9.2f ms' % ('$setting', '$1', '$2', d['value'], d['ms_per_step']))"
  done
done
