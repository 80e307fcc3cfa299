#!/bin/bash
# usage: tools_pmc.sh TAG WORKLOAD SPP   (runs on the GPU box; writes gpurun_out/pmc_TAG_*)
TAG=$1; W=$2; SPP=$3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_${TAG}_a -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmc_${TAG}_a.json 2> gpurun_out/pmc_${TAG}_a.err || exit 1
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_${TAG}_b -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmc_${TAG}_b.json 2> gpurun_out/pmc_${TAG}_b.err || exit 1
# HBM-side traffic, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2)
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_c -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmc_${TAG}_c.json 2> gpurun_out/pmc_${TAG}_c.err || exit 1
timeout -s KILL ${PMC_TIMEOUT:-120} rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_${TAG}_d -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-sub --workload $W --spp $SPP > gpurun_out/pmc_${TAG}_d.json 2> gpurun_out/pmc_${TAG}_d.err || exit 1
