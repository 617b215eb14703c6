#!/bin/bash
# Runs the rocprofv3 passes behind profiles/<TAG>_*: kernel trace + three separate PMC passes (never combined with tracing).
# usage (on the GPU box, from the repo root):  [BENCH_ARGS='--workload ecsample30x-like'] bash profiles/run_profiles.sh TAG [kt|all]
# then, back in the container:  python3 profiles/make_summary.py TAG gpurun_out/TAG_kt gpurun_out/TAG_fetch gpurun_out/TAG_write gpurun_out/TAG_sq
TAG=${1:?tag}
WHAT=${2:-all}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -- python3 $R/bench.py --steps 10 --warmup 2 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_kt.log 2>&1
if [ "$WHAT" = "all" ]; then
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/bench.py --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_write.log 2>&1
timeout 300 rocprofv3 --pmc TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum --output-format csv -d $R/gpurun_out/${TAG}_req -- python3 $R/bench.py --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_req.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/bench.py --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_sq.log 2>&1
fi
f=$(ls $R/gpurun_out/${TAG}_kt/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:40]:
    print("%-70s calls %5s avg_us %9.2f" % (r["Name"].replace("elba::(anonymous namespace)::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
