#!/bin/bash
# Round 4, final sources: the default bench line (cpu_baseline, loss section, training step), the other configurations with their
# cpu_baseline and PMC traffic (profiles/traffic.json of the same sources), B = 32, and the rocprofv3 kernel-trace stats of the
# default command.  bash profiles/r4_final_bench.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r4_bench.json 2> gpurun_out/r4_bench.err || { echo "default bench failed"; tail -5 gpurun_out/r4_bench.err; exit 1; }
echo "c2 done"
for cfg in c1 c1k8 c3 c4 c4n26 c5 c5f32; do
  timeout -k 10 400 python bench.py --config $cfg > gpurun_out/r4_${cfg}_bench.json 2> gpurun_out/r4_${cfg}_bench.err || { echo "$cfg failed"; tail -5 gpurun_out/r4_${cfg}_bench.err; exit 1; }
  echo "$cfg done"
done
timeout -k 10 400 python bench.py --batch 32 --steps 100 --no-train --no-section > gpurun_out/r4_b32_bench.json 2> gpurun_out/r4_b32_bench.err || { echo "b32 failed"; tail -5 gpurun_out/r4_b32_bench.err; exit 1; }
PEA_ZMARCH=0 timeout -k 10 400 python bench.py --config c4 --no-cpu-baseline > gpurun_out/r4_c4_zmarch0_bench.json 2> gpurun_out/r4_c4_zmarch0.err || echo "c4 zmarch0 failed"
# the same box with the producer / consumer f16 backward switched off
PEA_H16_HW=1 timeout -k 10 400 python bench.py --config c5 --no-cpu-baseline > gpurun_out/r4_c5_hw1_bench.json 2> gpurun_out/r4_c5_hw1.err || echo "c5 hw1 failed"
bash profiles/run_profile.sh r4 > gpurun_out/r4_profile.txt 2>&1
python3 - <<'PY'
import json
for k in ("bench", "c1_bench", "c1k8_bench", "c3_bench", "c4_bench", "c4_zmarch0_bench", "c4n26_bench", "c5_bench", "c5_hw1_bench", "c5f32_bench", "b32_bench"):
    try:
        j = json.loads(open("gpurun_out/r4_%s.json" % k).read().strip().splitlines()[-1])
    except Exception as ex:
        print(k, "missing", ex); continue
    r = j["roofline"]
    print(k, j["ms_per_step"], j.get("ms_per_step_autograd_seed"), j.get("graph_replay_ms"), j.get("kernel_ms"), r["frac"], r.get("fwd_plus_bwd_frac"), r.get("traffic"),
          (j.get("cpu_baseline") or {}).get("value"))
PY
