#!/bin/bash
# usage: tools/pmc_mem.sh TAG WORKLOAD SPP — texture-path counters (TA / TCP) of one bench step, per kernel
TAG=$1; W=$2; SPP=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcm_${TAG} -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmcm_${TAG}.json 2> gpurun_out/pmcm_${TAG}.err || exit 1
python3 - <<PY
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/pmcm_${TAG}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        m = re.search(r"rtk::(\w+)<false>", r["Kernel_Name"])
        if m: agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in agg.items():
    print(k, {n: "%.3g" % v for n, v in a.items()}, "TA busy share of (GUI_ACTIVE x 256 TAs/8 XCD-summed): %.1f%%" % (100 * a["TA_TA_BUSY_sum"] / (a["GRBM_GUI_ACTIVE"] / 8 * 256) if a["GRBM_GUI_ACTIVE"] else 0))
PY
