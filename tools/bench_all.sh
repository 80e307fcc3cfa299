#!/bin/bash
# usage: tools_bench_all.sh TAG  — runs the four workloads at reduced spp and prints one line each
TAG=$1
rm -f gpurun_out/bench_all_$TAG.jsonl
for w in "C1 0" "C2 200" "C3 32" "C4 8"; do set -- $w
  timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu --workload $1 --spp $2 >> gpurun_out/bench_all_$TAG.jsonl 2>> gpurun_out/bench_all_$TAG.err || exit 1
done
python3 - <<PY
import json
for l in open("gpurun_out/bench_all_$TAG.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:34], "Mrays/s %.0f"%d["value"], "Mpaths/s %.0f"%d["mpaths_per_s"], "ms %.2f"%d["roofline"]["kernel_ms"], "TF %.2f"%d["roofline"]["achieved"], "occ", d.get("phase_occupancy"), "runs/ray", d.get("phase_wave_runs_per_ray"))
PY
