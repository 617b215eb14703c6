#!/bin/bash
# one PMC pass restricted to the numeric kernel: bash profiles/tools/pmc_one.sh TAG COUNTER [COUNTER...]
TAG=${1:?tag}; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc "$@" --kernel-include-regex "k_spgemm_rows" --output-format csv -d $R/gpurun_out/${TAG} -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-align $BENCH_EXTRA > $R/gpurun_out/${TAG}.log 2>&1 || echo "pass failed: $@"
python3 - $R/gpurun_out/$TAG <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    for c in acc[k]: print("%-42s %-40s per launch %16.1f (%d)" % (k, c, acc[k][c] / n[k][c], n[k][c]))
PY
