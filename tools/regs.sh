#!/bin/bash
# usage: tools/regs.sh [extra hipcc flags] — VGPR / SGPR / LDS / scratch of every render kernel (device-only compile, no GPU needed)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -I$R/include -S --cuda-device-only "$@" -o /tmp/isa/cur.s $R/raytracing-in-one-weekend_amd/csrc/rt_device.hip 2>/dev/null
grep -E "^\s+\.(vgpr_count|sgpr_count|name|group_segment_fixed_size|private_segment_fixed_size):" /tmp/isa/cur.s | paste - - - - - | \
  sed -E 's/ +/ /g; s/_ZN3rtk[0-9]+//; s/EvNS_.*\.private/ .private/; s/EPK.*\.private/ .private/; s/ENS_.*\.private/ .private/' | \
  awk '{print $4, "lds", $2, "scratch", $6, "sgpr", $8, "vgpr", $10}' | sort
