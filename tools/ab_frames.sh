#!/bin/bash
# A/B of frames in flight (bench.py --frames) and hardware queues in alternating rounds.
# usage: bash tools/ab_frames.sh ROUNDS "C1 10 30|C3 100 2" "frames queues" ...
ROUNDS=$1; WLS=$2; shift 2
MODES=("$@")
for ((r = 0; r < ROUNDS; ++r)); do
  for mode in "${MODES[@]}"; do
    set -- $mode; fr=$1; hq=$2
    IFS='|' read -ra arr <<< "$WLS"
    for wl in "${arr[@]}"; do
      set -- $wl
      GPU_MAX_HW_QUEUES=$hq timeout -k 10 400 python3 bench.py --workload $1 --spp $2 --steps $3 --warmup 2 --frames $fr --no-cpu --no-sub > gpurun_out/ab_tmp.json 2> gpurun_out/ab_tmp.err || { echo "frames $fr $wl FAILED"; tail -5 gpurun_out/ab_tmp.err; exit 1; }
      python3 -c "
import json
d=json.loads(open('gpurun_out/ab_tmp.json').read().strip().splitlines()[-1])
print('round $r  frames $fr  hw queues %2d   %-6s %5s spp: %8.0f Mrays/s %9.3f ms/step  frame %.3f ms' % ($hq, '$1', '$2', d['value'], d['ms_per_step'], d['roofline']['frame_ms']))"
    done
  done
done
