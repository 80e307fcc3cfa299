#!/bin/bash
# usage: tools/ktrace.sh TAG WORKLOAD SPP [extra bench.py flags] — rocprofv3 kernel-trace stats of one bench run (on the GPU box)
TAG=$1; W=$2; SPP=$3; shift 3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
timeout -s KILL ${KT_TIMEOUT:-300} rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt_${TAG} -- python3 bench.py --steps 2 --warmup 1 --no-cpu --workload $W --spp $SPP "$@" > gpurun_out/kt_${TAG}.json 2> gpurun_out/kt_${TAG}.err || exit 1
f=$(ls gpurun_out/kt_${TAG}/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/kt_${TAG}_kernel_stats.csv
head -24 $f | cut -d, -f1-8
