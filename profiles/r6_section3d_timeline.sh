#!/bin/bash
# bash profiles/r6_section3d_timeline.sh: rocprofv3 --kernel-trace of profiles/r6_section3d.py (one_node, composed), and per form the
# timeline of ONE call (the last of the drained ones): kernel, queue, start and end relative to the call's first kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for W in one_node finished composed; do
  OUT=$ROOT/gpurun_out/r6_sec3d_$W; rm -rf $OUT; mkdir -p $OUT
  timeout -k 5 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o r -- python3 $ROOT/profiles/r6_section3d.py $W 6 > $OUT/log.txt 2>&1 || echo "FAILED/timeout"
  tail -2 $OUT/log.txt
  python3 - $OUT $W <<'PY'
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# bursts: a gap of > 150 us between one kernel's end and the next one's start separates the calls (each drained by a synchronize)
bursts, cur = [], []
for r in rows:
    if cur and int(r["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in cur) > 150000:
        bursts.append(cur); cur = []
    cur.append(r)
bursts.append(cur)
# the last single-call burst before the timed batch of ten (the batch is the longest burst)
singles = [b for b in bursts if len(b) < max(len(x) for x in bursts)]
b = singles[-1]
t0 = int(b[0]["Start_Timestamp"])
print("== %s: one call, %d kernels, %.1f us from the first kernel's start to the last one's end" % (sys.argv[2], len(b), (max(int(x["End_Timestamp"]) for x in b) - t0) / 1e3))
print("%-8s %9s %9s %8s  %s" % ("queue", "start us", "end us", "dur us", "kernel (grid)"))
for r in b:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%-8s %9.1f %9.1f %8.1f  %s (%s)" % (r.get("Queue_Id", "?"), s, e, e - s, r["Kernel_Name"].replace("void ", "").replace("pea::", "")[:110], r.get("Grid_Size_X", r.get("Grid_Size", "?"))))
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in b) / 1e3
print("sum of kernel durations %.1f us" % busy)
PY
done
