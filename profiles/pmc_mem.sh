#!/bin/bash
# bash profiles/pmc_mem.sh <tag> <fwd|bwd|inf|fused>   memory-pipe counters (TA / TCP) -> gpurun_out/pmcm_<tag>/summary.txt
TAG=$1; WHICH=$2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcm_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "TA_BUSY_sum TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_BUFFER_COALESCED_READ_CYCLES_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/g$i -o r -- python3 $ROOT/profiles/one_kernel.py $WHICH 3 > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python3 - $OUT <<'PY' | tee $OUT/summary.txt
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pea::", "").split("(")[0][:70]
        if not k.startswith("k_") or "finalize" in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
for k, cs in acc.items():
    print("==", k)
    for c in sorted(cs):
        v = cs[c]
        print("   %-40s %14.5g" % (c, sum(v) / len(v)))
PY
