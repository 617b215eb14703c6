#!/bin/bash
# kernel trace of the SpGEMM steps of a workload (the k-mer stage's kernels filtered out by name), per-kernel averages and the timeline of ONE step:
# usage (GPU box, repo root): bash profiles/tools/kt_step.sh WORKLOAD [bench args]
W=$1; shift
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
d=$R/gpurun_out/kts_$W
rm -rf $d
rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $R/bench.py --workload $W --steps 6 --warmup 2 --steady-steps 0 --no-cpu-baseline --no-accounting "$@" > $d.log 2>&1
f=$(ls $d/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"].replace("elba::(anonymous namespace)::", "").replace("elba::", "")[:70] for r in rows]
# the last step: from the last k_zero_regions (or k_classify) to the end
last = max(i for i, n in enumerate(names) if n.startswith("k_zero_regions") or n.startswith("k_classify"))
firsts = [i for i, n in enumerate(names) if n.startswith("k_zero_regions")]
lo = firsts[-1] if firsts else last
t0 = int(rows[lo]["Start_Timestamp"]); prev_end = t0
for i in range(lo, len(rows)):
    s, e = int(rows[i]["Start_Timestamp"]), int(rows[i]["End_Timestamp"])
    print("%9.1f us  +gap %6.1f  dur %8.1f  grid %8s wg %5s  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, rows[i].get("Grid_Size_X", rows[i].get("Grid_Size", "?")), rows[i].get("Workgroup_Size_X", rows[i].get("Workgroup_Size", "?")), names[i]))
    prev_end = e
print("step span us %.1f" % ((prev_end - t0) / 1e3))
PY
