#!/bin/bash
for ls in 1 2 3 4; do echo "== RT_SUB_LOG2=$ls"
for w in "C1 0" "C2 1000" "C3 128" "C4 64"; do set -- $w
  RT_SUB_LOG2=$ls timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --no-cpu --workload $1 --spp $2 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['config']['workload'][:40], 'Mrays/s %.0f'%d['value'], 'ms %.2f'%d['roofline']['kernel_ms'])"
done; done
