#!/bin/bash
# usage: tools/bench_big.sh TAG — the four workloads at sizes where the pool is refilled many times
TAG=$1
rm -f gpurun_out/bench_big_$TAG.jsonl
for w in "C1 100" "C2 1000" "C3 128" "C4 32" "GP 256" "BLOCKS 256"; do set -- $w
  timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu --workload $1 --spp $2 >> gpurun_out/bench_big_$TAG.jsonl 2>> gpurun_out/bench_big_$TAG.err || exit 1
done
python3 - <<PY
import json
for l in open("gpurun_out/bench_big_$TAG.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:40], "Mrays/s %.0f"%d["value"], "ms %.2f"%d["roofline"]["kernel_ms"], "occ", [d["phase_occupancy"][k] for k in ("node_step","leaf_step","begin_trace")])
PY
