#!/bin/bash
# Round 6, final sources, part b: every configuration on its own with its cpu_baseline.  bash profiles/r6_final_bench_b.sh [configs...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
CFGS=${@:-c1 c1k8 c3 c3crop c4 c4crop c4n26 c5 c5f32 c2ema c3ema c4ema c5ema c4r6}
for cfg in $CFGS; do
  timeout -k 10 400 python bench.py --config $cfg > gpurun_out/r6_${cfg}_bench.json 2> gpurun_out/r6_${cfg}_bench.err || { echo "$cfg failed"; tail -5 gpurun_out/r6_${cfg}_bench.err; exit 1; }
  python3 - $cfg <<'PY'
import json, sys
j = json.loads(open("gpurun_out/r6_%s_bench.json" % sys.argv[1]).read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], j["ms_per_step"], j.get("ms_min"), j.get("ms_max"), j.get("graphed_api_ms"), j.get("kernel_ms"), r["frac"], r.get("fwd_plus_bwd_frac"), r.get("traffic"),
      (j.get("cpu_baseline") or {}).get("value"), (j.get("cpu_baseline") or {}).get("best_cpu_value"))
PY
done
