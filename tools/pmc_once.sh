#!/bin/bash
# usage: tools/pmc_once.sh TAG "COUNTERS" WORKLOAD SPP   — one rocprofv3 --pmc pass of one bench step; prints per-kernel sums
TAG=$1; CTRS=$2; W=$3; SPP=$4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc $CTRS --output-format csv -d gpurun_out/pmc1_${TAG} -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmc1_${TAG}.json 2> gpurun_out/pmc1_${TAG}.err || { tail -5 gpurun_out/pmc1_${TAG}.err; exit 1; }
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob("gpurun_out/pmc1_${TAG}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"rtk::(\w+)<false", r["Kernel_Name"])
        if not m: continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[m.group(1)] += 1
for k, a in agg.items():
    print(k, "dispatches", n[k], {c: ("%.4g" % v) for c, v in sorted(a.items())})
PY
