#!/bin/bash
for f in gpurun_variants/lib_*.so; do echo "== $f"; RTIOW_HIP_LIB=$PWD/$f ./tools/bench_big.sh var | tail -4; done
