#!/bin/bash
# PMC survey of the numeric kernel: several separate counter passes (never combined with tracing), condensed per kernel name.
# usage (GPU box, repo root): bash profiles/tools/pmc_survey.sh TAG
TAG=${1:?tag}
R=$(pwd)
cd /tmp && export TMPDIR=/tmp
i=0
while read -r grp; do
  [ -z "$grp" ] && continue
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/${TAG}_pmc$i -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-align > $R/gpurun_out/${TAG}_pmc$i.log 2>&1 || echo "group $i failed: $grp"
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU
SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM_RD
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_ATOMIC_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_CC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TAGRAM0_REQ_sum
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
    if "spgemm_rows" not in k and "finalize" not in k and "mirror" not in k: continue
    print(k)
    for c in sorted(acc[k]): print("   %-42s per launch %16.1f   (launches %d)" % (c, acc[k][c] / max(1, calls[k][c]), calls[k][c]))
PY
