#!/bin/bash
# usage (GPU box, repo root): bash profiles/tools/ab_kmer.sh REPEATS name1 name2 ...   ("base" = the in-tree build; "base:opt=1" sets an elba_set_option through ELBA_BENCH_OPTIONS)
# k-mer stage phases of bench.py's default workload, interleaved runs
N=$1; shift
R=$(cd $(dirname $0)/../.. && pwd)
for r in $(seq $N); do
  for v in "$@"; do
    lib=${v%%:*}; opts=""; [ "$lib" != "$v" ] && opts=${v#*:}
    if [ "$lib" = "base" ]; then L=""; else L="$R/scratch/variants/$lib/libelba_amd.so"; fi
    ELBA_BENCH_OPTIONS=$opts ELBA_AMD_LIB=$L python $R/bench.py --no-cpu-baseline --no-accounting --steps 5 --warmup 1 --steady-steps 0 $BENCH_ARGS 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j['kmer_stage']
print('$v', 'kmer', k['wall_ms'], 'partition', k['count_ms'], 'buckets', k['runs_to_columns_ms'], 'csr', k['matrix_build_ms'], '| step', j['ms_per_step'], 'numeric', j['phases_ms']['ms_numeric'])
"
  done
done
