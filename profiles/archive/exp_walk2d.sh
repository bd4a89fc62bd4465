#!/bin/bash
# bash profiles/exp_walk2d.sh: the 2D tile walk (PEA_WALK2D = strip width in tiles, 0 = row-major) on c3 (D=32, 704^2) and c2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
for cfg in c3 c2 c5; do
for w in 0 4 6 8 11; do
  PEA_WALK2D=$w timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-train --no-section > gpurun_out/walk_${cfg}_$w.json 2> gpurun_out/walk_${cfg}_$w.err || { echo "$cfg $w failed"; tail -3 gpurun_out/walk_${cfg}_$w.err; }
  python3 - $cfg $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/walk_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], "walk2d", sys.argv[2], "ms_per_step", j["ms_per_step"], {k: j["kernel_ms"][k] for k in ("fwd", "bwd")}, flush=True)
PY
done
done
