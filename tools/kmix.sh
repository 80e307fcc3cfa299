#!/bin/bash
# usage: tools/kmix.sh MANGLED_PREFIX [file.s] — memory-instruction mix and VALU count of one kernel in the ISA listing (tools/regs.sh writes /tmp/isa/cur.s)
F=${2:-/tmp/isa/cur.s}
awk -v k="^$1" '$0 ~ k && /:/ {f=1} f && /\.Lfunc_end/ {exit} f {print}' $F | awk '{print $1}' | grep -E "^(ds_|global_|flat_|scratch_|buffer_|v_|s_)" | \
  awk '{c[$1]++} END{for(k in c) if (k ~ /^(ds_|global_|flat_|scratch_|buffer_)/) printf "%s=%d ", k, c[k]; n=0; m=0; for(k in c) {if (k ~ /^v_/) n+=c[k]; if (k ~ /^s_/) m+=c[k]} print "| valu", n, "salu", m}'
