#!/bin/bash
# Round 3, final sources: the default bench line (cpu_baseline, loss section, training step), the other configurations with their
# cpu_baseline and PMC traffic (profiles/traffic.json of the same sources), and the rocprofv3 kernel-trace stats of the default command.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r3_bench.json 2> gpurun_out/r3_bench.err || { echo "default bench failed"; tail -5 gpurun_out/r3_bench.err; exit 1; }
echo "c2 done"
for cfg in c3 c4 c4n26 c5 c5f32; do
  timeout -k 10 400 python bench.py --config $cfg > gpurun_out/r3_${cfg}_bench.json 2> gpurun_out/r3_${cfg}_bench.err || { echo "$cfg failed"; tail -5 gpurun_out/r3_${cfg}_bench.err; exit 1; }
  echo "$cfg done"
done
bash profiles/run_profile.sh r3 > gpurun_out/r3_profile.txt 2>&1
python3 - <<'PY'
import json
for k in ("bench", "c3_bench", "c4_bench", "c4n26_bench", "c5_bench", "c5f32_bench"):
    j = json.loads(open("gpurun_out/r3_%s.json" % k).read().strip().splitlines()[-1])
    r = j["roofline"]
    print(k, j["ms_per_step"], j.get("kernel_ms"), r["frac"], r.get("fwd_plus_bwd_frac"), r.get("traffic"), (j.get("cpu_baseline") or {}).get("value"))
PY
