#!/bin/bash
# usage (GPU box, repo root): [BENCH_ARGS=...] bash profiles/tools/ab.sh REPEATS name1 name2 ...   ("base" = the in-tree build); interleaved runs
N=$1; shift
R=$(cd $(dirname $0)/../.. && pwd)
for r in $(seq $N); do
  for v in "$@"; do
    if [ "$v" = "base" ]; then L=""; else L="$R/scratch/variants/$v/libelba_amd.so"; fi
    ELBA_AMD_LIB=$L python $R/bench.py --no-cpu-baseline --no-accounting --steps 20 --steady-steps 0 $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'ms_step', j['ms_per_step'], 'numeric', j['phases_ms']['ms_numeric'], 'fin', j['phases_ms']['ms_finalize'], 'dev', j['phases_ms']['ms_total'], 'kmer', j['kmer_stage']['wall_ms'], 'mirror', j.get('mirror'))
"
  done
done
