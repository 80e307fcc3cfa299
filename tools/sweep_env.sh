#!/bin/bash
# usage: tools/sweep_env.sh VAR "v1 v2 ..." WORKLOAD SPP [steps] — one bench line per value of an environment knob (on the GPU box)
VAR=$1; VALS=$2; W=$3; SPP=$4; STEPS=${5:-3}
for v in $VALS; do
  if [ "$v" = "default" ]; then unset $VAR; else export $VAR=$v; fi
  timeout -k 10 300 python3 bench.py --no-cpu --no-sub --workload $W --spp $SPP --steps $STEPS --warmup 1 > gpurun_out/sweep_tmp.json 2> gpurun_out/sweep_tmp.err || { echo "$VAR=$v FAILED"; tail -3 gpurun_out/sweep_tmp.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/sweep_tmp.json').read().strip().splitlines()[-1])
print('$VAR=$v', '$W', '$SPP', 'spp: %.0f Mrays/s  %.2f ms' % (d['value'], d['ms_per_step']))"
done
