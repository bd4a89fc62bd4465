#!/bin/bash
# bash profiles/exp_lab.sh: the labels-in forward (k_fwd_xdma<.., LAB>) timed by rocprofv3 with and without the interior-tile store walk
# (the walk and its PEA_LAB_SLOW switch existed in that experiment's build only: profiles/r4_lab_interior.txt; kept as the recipe)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for slow in "" 1 "" 1; do
  OUT=$ROOT/gpurun_out/lab_${slow:-0}; rm -rf $OUT; mkdir -p $OUT
  if [ -n "$slow" ]; then export PEA_LAB_SLOW=1; else unset PEA_LAB_SLOW; fi
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o r -- python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-section > $OUT/log.txt 2>&1 || echo failed
  python3 - $OUT "${slow:-0}" <<'PY'
import csv, glob, os, sys
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "true, 0, false, 6, true" in r["Name"] or "k_bwd_xdma<16" in r["Name"]:
            print("slow=%s" % sys.argv[2], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3), r["Name"][:90])
PY
done
