R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04_acc_kt -- python3 $R/bench.py --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline > $R/gpurun_out/r04_acc_kt.log 2>&1
f=$(ls $R/gpurun_out/r04_acc_kt/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:45]:
    print("%-70s calls %5s avg_us %9.2f tot_ms %8.2f" % (r["Name"].replace("elba::(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"])/1e6))
PY
