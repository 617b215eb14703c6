#!/bin/bash
for d in "$@"; do
  python bench.py --no-cpu-baseline --no-align --steps 100 --dbg $d 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dbg=$d', 'numeric', j['phases_ms']['ms_numeric'], 'fin', j['phases_ms']['ms_finalize'], 'total_dev', j['phases_ms']['ms_total'])
"
done
