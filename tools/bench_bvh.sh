#!/bin/bash
# usage: tools/bench_bvh.sh TAG — host-built (binned SAH) against device-built (Morton / Karras) BVH: build time and render rate
TAG=$1
rm -f gpurun_out/bench_bvh_$TAG.jsonl
for w in "C1 100" "C3 128" "C4 32"; do set -- $w
  for mode in "" "--device-bvh=ploc" "--device-bvh=lbvh"; do
    timeout -k 10 300 python -u bench.py --steps 2 --warmup 1 --no-cpu --workload $1 --spp $2 $mode >> gpurun_out/bench_bvh_$TAG.jsonl 2>> gpurun_out/bench_bvh_$TAG.err || exit 1
  done
done
python3 - <<PY
import json
for l in open("gpurun_out/bench_bvh_$TAG.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:32], "| Mrays/s %.0f"%d["value"], "| nodes", d["config"]["bvh_nodes"], "|", d["config"]["bvh"], "| node visits/ray %.1f" % d.get("node_visits_per_ray", -1))
PY
