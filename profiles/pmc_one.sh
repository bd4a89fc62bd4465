#!/bin/bash
# bash profiles/pmc_one.sh <tag> <fwd|bwd|inf|fused> "<counters>"   one PMC pass with a hard timeout -> prints per-kernel averages
TAG=$1; WHICH=$2; CTRS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc1_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 90 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o r -- python3 $ROOT/profiles/one_kernel.py $WHICH 3 > $OUT/log.txt 2>&1 || echo "FAILED/timeout: $CTRS"
python3 - $OUT $TAG <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pea::", "").split("(")[0][:40]
        if not k.startswith("k_") or "finalize" in k: continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    print(sys.argv[2], k, {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())})
PY
