#!/bin/bash
# the two SQ counter passes of pmc_kmer.sh on kernels matching REGEX, through kmer_only.py: bash profiles/tools/pmc_sq.sh TAG REGEX [kmer_only args]
TAG=${1:?tag}; RX=${2:?regex}; shift 2
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 280 rocprofv3 --pmc $grp --kernel-include-regex "$RX" --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/profiles/tools/kmer_only.py "$@" > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS
SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC
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
