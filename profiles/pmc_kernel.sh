#!/bin/bash
# bash profiles/pmc_kernel.sh <tag> <fwd|bwd|inf>   (PEA_* env passes through) -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; WHICH=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python3 $ROOT/profiles/one_kernel.py $WHICH 4 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pea::", "").split("(")[0][:70]
        if not k.startswith("k_"): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        acc[k]["_vgpr"] = [float(r["VGPR_Count"])]; acc[k]["_lds"] = [float(r["LDS_Block_Size"])]; acc[k]["_scratch"] = [float(r["Scratch_Size"])]
for k, cs in acc.items():
    print("==", k)
    for c in sorted(cs):
        v = cs[c]
        print("   %-26s %14.4g" % (c, sum(v) / len(v)))
PY
