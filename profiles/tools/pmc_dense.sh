#!/bin/bash
# PMC passes (never combined with tracing) restricted to the numeric kernel of the dense path, bench.py --workload dense-repeats-25th; per launch, per kernel name:
# usage (GPU box, repo root): bash profiles/tools/pmc_dense.sh TAG
TAG=${1:?tag}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "k_spgemm_direct" --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/bench.py --workload dense-repeats-25th --steps 3 --warmup 1 --steady-steps 0 --no-cpu-baseline --no-accounting > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GROUPS'
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT
GROUPS
python3 - $R/gpurun_out $TAG <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("%s/%s_pmc*/*/*counter_collection.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("elba::(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c in sorted(acc[k]): print("   %-28s per launch %14.4g   (launches %d)" % (c, acc[k][c] / max(1, calls[k][c]), calls[k][c]))
    a = acc[k]; n = lambda c: a[c] / max(1, calls[k][c])
    if "SQ_LDS_IDX_ACTIVE" in a and n("SQ_LDS_IDX_ACTIVE"): print("   => SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = %.3f" % (n("SQ_LDS_BANK_CONFLICT") / n("SQ_LDS_IDX_ACTIVE")))
    if "TCC_HIT_sum" in a and (n("TCC_HIT_sum") + n("TCC_MISS_sum")): print("   => TCC hit rate = %.3f" % (n("TCC_HIT_sum") / (n("TCC_HIT_sum") + n("TCC_MISS_sum"))))
PY
