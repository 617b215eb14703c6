#!/bin/bash
# PMC survey of the k-mer stage's kernels: separate counter passes (never combined with tracing), condensed per kernel name.
# usage (GPU box, repo root): [BENCH_ARGS=...] bash profiles/tools/pmc_kmer.sh TAG
TAG=${1:?tag}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "k_msd|k_rs_|k_unpack|k_cs_" --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/bench.py --steps 1 --warmup 0 --steady-steps 0 --no-cpu-baseline --no-accounting $BENCH_ARGS > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
FETCH_SIZE
WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
TCC_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
GROUPS
python3 - $R/gpurun_out $TAG <<'PY'
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob("%s/%s_pmc*/*/*counter_collection.csv" % (root, tag)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("elba::(anonymous namespace)::", "").split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]): print("   %-42s per launch %18.1f   (launches %d)" % (c, acc[k][c] / max(1, calls[k][c]), calls[k][c]))
PY
