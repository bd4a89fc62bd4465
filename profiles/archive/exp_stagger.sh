#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
for cfg in c2 c3 c5; do
for w in 0 1 0 1; do
  PEA_XCD_STAGGER=$w timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline --no-train --no-section > gpurun_out/stg_${cfg}_$w.json 2> gpurun_out/stg_${cfg}_$w.err || { echo "$cfg $w failed"; tail -3 gpurun_out/stg_${cfg}_$w.err; }
  python3 - $cfg $w <<'PY'
import json, sys
j = json.loads(open("gpurun_out/stg_%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print(sys.argv[1], "stagger", sys.argv[2], "ms_per_step", j["ms_per_step"], {k: j["kernel_ms"][k] for k in ("fwd", "bwd")}, flush=True)
PY
done
done
PEA_XCD_STAGGER=1 timeout -k 10 600 python -m pytest tests/test_gpu_cross.py -x -q -m gpu 2>&1 | tail -3
