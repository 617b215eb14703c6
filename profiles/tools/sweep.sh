#!/bin/bash
# usage: bash scratch/sweep.sh VAR v1 v2 ...   (prints ms_per_step / numeric ms per setting)
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --no-cpu-baseline --no-align --steps 300 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$VAR=$v', 'ms_step', j['ms_per_step'], 'numeric', j['phases_ms']['ms_numeric'], 'fin', j['phases_ms']['ms_finalize'], 'total_dev', j['phases_ms']['ms_total'], j['tiers'])
"
done
