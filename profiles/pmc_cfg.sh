#!/bin/bash
# bash profiles/pmc_cfg.sh <tag> <config> [passes...]: PMC passes over `bench.py --config <config>` (few steps), per-kernel averages
# (PEA_BENCH_EXTRA: more bench.py arguments, e.g. "--batch 32")
TAG=$1; CFG=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  OUT=$ROOT/gpurun_out/pmcc_${TAG}_$1; rm -rf $OUT; mkdir -p $OUT
  timeout -k 5 200 rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT -o r -- python3 $ROOT/bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-train --no-section $PEA_BENCH_EXTRA > $OUT/log.txt 2>&1 || echo "FAILED/timeout: $2"
  python3 - $OUT $TAG <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").replace("pea::", "").split("(")[0][:30]
        if not (k.startswith("k_bwd") or k.startswith("k_fwd")): continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc[k]["dur_us"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
for k, cs in acc.items():
    print(sys.argv[2], k, {c: round(sum(v[-4:]) / len(v[-4:]), 1) for c, v in sorted(cs.items())})
PY
}
for p in "$@"; do
case $p in
a) run a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU";;
b) run b "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS";;
c) run c "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum";;
f) run f "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum";;
g) run g "FETCH_SIZE";;
h) run h "WRITE_SIZE";;
esac
done
