#!/bin/bash
# usage: bash scratch/ab.sh REPEATS name1 name2 ...   ("base" = the in-tree build); interleaved runs, numeric / total device ms per run
N=$1; shift
R=$(cd $(dirname $0)/.. && pwd)
for r in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "base" ]; then L=""; else L="$R/scratch/variants/$v/libelba_amd.so"; fi
    ELBA_AMD_LIB=$L python $R/bench.py --no-cpu-baseline --no-align --steps 300 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms_step', j['ms_per_step'], 'numeric', j['phases_ms']['ms_numeric'], 'fin', j['phases_ms']['ms_finalize'], 'dev', j['phases_ms']['ms_total'], 'parity', j['parity_vs_oracle'])
"
  done
done
