#!/bin/bash
# Round 6, final sources: the default bench line (cpu_baseline, loss sections, every BASELINE config in `configs`, training step), every
# configuration on its own with its cpu_baseline (c4crop = the reference's 3D training crops is new this round), B = 32, and the rocprofv3
# kernel-trace stats of the default command.  bash profiles/r6_final_bench_a.sh (part a: the default lines, B = 32, the rocprofv3 stats; part b: every configuration on its own)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err || { echo "default bench failed"; tail -5 gpurun_out/r6_bench.err; exit 1; }
echo "c2 done"
# the driver's own command line (steps 20, warm-up 5), twice: what its box-to-box spread looks like inside one box
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r6_bench_s20_$i.json 2> gpurun_out/r6_bench_s20_$i.err || echo "s20 $i failed"; done
timeout -k 10 400 python bench.py --batch 32 --steps 100 --no-train --no-section > gpurun_out/r6_b32_bench.json 2> gpurun_out/r6_b32_bench.err || { echo "b32 failed"; tail -5 gpurun_out/r6_b32_bench.err; exit 1; }
bash profiles/run_profile.sh r6 > gpurun_out/r6_profile.txt 2>&1
python3 - <<'PY'
import json
for k in ("bench", "bench_s20_1", "bench_s20_2", 
          "b32_bench"):
    try:
        j = json.loads(open("gpurun_out/r6_%s.json" % k).read().strip().splitlines()[-1])
    except Exception as ex:
        print(k, "missing", ex); continue
    r = j["roofline"]
    print(k, j["ms_per_step"], j.get("ms_min"), j.get("ms_max"), j.get("graph_replay_ms"), j.get("graph_replay_x8_ms"), j.get("graphed_api_ms"), j.get("kernel_ms"),
          r["frac"], r.get("fwd_plus_bwd_frac"), r.get("traffic"), (j.get("cpu_baseline") or {}).get("value"), j.get("loss_section_us"))
PY
