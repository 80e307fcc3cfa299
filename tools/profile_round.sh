#!/bin/bash
# usage: tools/profile_round.sh "C1 10|C2 1000|C3 10000|C4 256"  (on the GPU box) — for each workload at its full size: rocprofv3 kernel-trace
# stats and the four PMC passes (SQ a, SQ b, FETCH_SIZE, WRITE_SIZE) of one bench step each, summarised into gpurun_out/prof_<wl>_<spp>spp_*
export PMC_TIMEOUT=500 KT_TIMEOUT=500
IFS='|' read -ra arr <<< "$1"
for wl in "${arr[@]}"; do
  set -- $wl
  n=$(echo $1 | tr A-Z a-z)
  tag=${n}_$2
  bash tools/ktrace.sh $tag $1 $2 --no-sub --frames 1 > gpurun_out/prof_${n}_${2}spp_ktrace.txt 2>&1 || { echo "ktrace $wl failed"; tail -3 gpurun_out/kt_$tag.err; exit 1; }
  cp gpurun_out/kt_${tag}_kernel_stats.csv gpurun_out/prof_${n}_${2}spp_kernel_stats.csv
  cp gpurun_out/kt_${tag}.json gpurun_out/prof_${n}_${2}spp_ktrace_bench.json
  bash tools/pmc.sh $tag $1 $2 || { echo "pmc $wl failed"; exit 1; }
  python3 tools/pmc_report.py $tag gpurun_out/prof_${n}_${2}spp_pmc.json > gpurun_out/prof_${n}_${2}spp_pmc.txt || exit 1
  rm -rf gpurun_out/kt_$tag gpurun_out/pmc_${tag}_?
  echo "done $wl"
done
