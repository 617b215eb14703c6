#!/bin/bash
# usage (GPU box, repo root): bash profiles/tools/kt_variants.sh FILTER name1 name2 ...   — rocprofv3 kernel trace of profiles/tools/kmer_only.py per build
# ("base" = in-tree, "name:opt=1,opt2=3" sets options), average time of the kernels matching FILTER
F=$1; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=${v%%:*}; opts=""; [ "$lib" != "$v" ] && opts=${v#*:}
  if [ "$lib" = "base" ]; then L=""; else L="$R/scratch/variants/$lib/libelba_amd.so"; fi
  d=$R/gpurun_out/ktv_$(echo $v | tr ':=,' '___')
  rm -rf $d
  ELBA_BENCH_OPTIONS=$opts ELBA_AMD_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/profiles/tools/kmer_only.py $KMER_ONLY_ARGS > $d.log 2>&1
  f=$(ls $d/*/*kernel_stats.csv | head -1)
  echo "== $v"; grep "^pass 1" $d.log
  python3 - "$f" "$F" <<'PY'
import csv, sys, re
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows:
    n = r["Name"].replace("elba::(anonymous namespace)::", "")
    if re.search(sys.argv[2], n): print("   %-60s calls %4s avg_us %9.2f" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
