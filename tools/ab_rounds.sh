#!/bin/bash
# A/B of environment settings in alternating rounds: every setting once per round, ROUNDS rounds, so that drift of the box
# (clock, neighbours) spreads over all of them.  usage: bash tools/ab_rounds.sh ROUNDS "C1 10 30|C1 100 5" "A=0 B=1" "A=1 B=1" ...
ROUNDS=$1; WLS=$2; shift 2
SETTINGS=("$@")
for ((r = 0; r < ROUNDS; ++r)); do
  for setting in "${SETTINGS[@]}"; do
    IFS='|' read -ra arr <<< "$WLS"
    for wl in "${arr[@]}"; do
      set -- $wl
      env RT_KNOBS=1 $setting timeout -k 10 300 python3 bench.py --workload $1 --spp $2 --steps $3 --warmup 1 --no-cpu --no-sub > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "$setting $wl FAILED"; tail -3 gpurun_out/ab_tmp.err; exit 1; }
      python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('round $r  %-52s %-3s %5s spp: %8.0f Mrays/s %9.3f ms' % ('$setting', '$1', '$2', d['value'], d['ms_per_step']))"
    done
  done
done
