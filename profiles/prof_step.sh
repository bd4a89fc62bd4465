#!/bin/bash
# bash profiles/prof_step.sh <tag>: kernel trace + stats of the bench step loop only (profiles/host_overhead.py)
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/step_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PEA_NO_TINY=1
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python3 $ROOT/profiles/host_overhead.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:14]:
        print("%-70s calls %6s avg %9.1f ns  %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]), r["Percentage"][:5]))
PY
