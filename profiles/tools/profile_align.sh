#!/bin/bash
# kernel trace of one bench run INCLUDING the alignment stage; prints the x-drop kernels' times
TAG=${1:?tag}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_aln -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_aln.log 2>&1
f=$(ls -t $R/gpurun_out/${TAG}_aln/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/${TAG}_align_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-80s calls %5s avg_ms %10.3f total_ms %10.3f" % (r["Name"].replace("elba::(anonymous namespace)::", "")[:80], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 $R/gpurun_out/${TAG}_aln.log | python3 -c "import sys, json; print(json.dumps(json.loads(sys.stdin.read())['align_stage']))"
