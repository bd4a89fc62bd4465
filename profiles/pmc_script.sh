#!/bin/bash
# bash profiles/pmc_script.sh <tag> <script.py> "<counters>": one PMC pass over a script; prints per-kernel averages of runs of
# consecutive dispatches of the same kernel (so a script that times several cases in sequence gives one row per case and kernel)
TAG=$1; SCRIPT=$2; CTRS=$3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcx_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 280 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT -o r -- python3 $ROOT/$SCRIPT > $OUT/log.txt 2>&1 || echo "FAILED/timeout"
python3 - $OUT <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = {}
for r in rows:
    d = disp.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"].replace("void ", "").replace("pea::", "").split("(")[0][:34],
                                                 "t": (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
runs = []
for i in sorted(disp):
    d = disp[i]
    if not (d["k"].startswith("k_fwd") or d["k"].startswith("k_bwd")): continue
    if runs and runs[-1][0]["k"] == d["k"] and abs(runs[-1][-1]["t"] - d["t"]) < 0.3 * d["t"]: runs[-1].append(d)
    else: runs.append([d])
for run in runs:
    keys = [k for k in run[0] if k not in ("k",)]
    print(run[0]["k"], len(run), {k: round(sum(d[k] for d in run) / len(run), 1) for k in keys})
PY
