#!/bin/bash
# bash profiles/prof_script.sh <tag> <script.py> [args]: rocprofv3 kernel trace + stats of one script; top kernels by total time
TAG=$1; SCRIPT=$2; shift; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/profs_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python3 $ROOT/$SCRIPT "$@" > $OUT/log.txt 2>&1 || echo "FAILED/timeout"
python3 - $OUT <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:24]:
        print("%-86s calls %6s avg %9.1f us  %5s%%" % (r["Name"].replace("void ", "").replace("pea::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"][:5]))
PY
tail -12 $OUT/log.txt
