#!/bin/bash
# Round 5, final sources: the default bench line (cpu_baseline, loss section, training step), every other configuration (the EMA cross
# losses and the BBBC training crops are new this round) with its cpu_baseline, B = 32, the tile-walk comparison line, and the rocprofv3
# kernel-trace stats of the default command.  bash profiles/r5_final_bench.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout -k 10 600 python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { echo "default bench failed"; tail -5 gpurun_out/r5_bench.err; exit 1; }
echo "c2 done"
# the driver's own command line (steps 20, warm-up 5), twice: what its box-to-box spread looks like inside one box
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r5_bench_s20_$i.json 2> gpurun_out/r5_bench_s20_$i.err || echo "s20 $i failed"; done
for cfg in c1 c1k8 c3 c3crop c4 c4n26 c5 c5f32 c2ema c3ema c4ema c5ema c4r6; do
  timeout -k 10 400 python bench.py --config $cfg > gpurun_out/r5_${cfg}_bench.json 2> gpurun_out/r5_${cfg}_bench.err || { echo "$cfg failed"; tail -5 gpurun_out/r5_${cfg}_bench.err; exit 1; }
  echo "$cfg done"
done
timeout -k 10 400 python bench.py --batch 32 --steps 100 --no-train --no-section > gpurun_out/r5_b32_bench.json 2> gpurun_out/r5_b32_bench.err || { echo "b32 failed"; tail -5 gpurun_out/r5_b32_bench.err; exit 1; }
# the same box with the backward's tile walk first tile first (round 4's order)
PEA_BWD_REV=0 timeout -k 10 400 python bench.py --no-cpu-baseline --no-train > gpurun_out/r5_bench_rev0.json 2> gpurun_out/r5_bench_rev0.err || echo "rev0 failed"
bash profiles/run_profile.sh r5 > gpurun_out/r5_profile.txt 2>&1
python3 - <<'PY'
import json
for k in ("bench", "bench_s20_1", "bench_s20_2", "bench_rev0", "c1_bench", "c1k8_bench", "c3_bench", "c3crop_bench", "c4_bench", "c4n26_bench", "c5_bench", "c5f32_bench",
          "c2ema_bench", "c3ema_bench", "c4ema_bench", "c5ema_bench", "c4r6_bench", "b32_bench"):
    try:
        j = json.loads(open("gpurun_out/r5_%s.json" % k).read().strip().splitlines()[-1])
    except Exception as ex:
        print(k, "missing", ex); continue
    r = j["roofline"]
    print(k, j["ms_per_step"], j.get("ms_min"), j.get("ms_max"), j.get("graph_replay_ms"), j.get("graph_replay_x8_ms"), j.get("graphed_api_ms"), j.get("kernel_ms"),
          r["frac"], r.get("fwd_plus_bwd_frac"), r.get("traffic"), (j.get("cpu_baseline") or {}).get("value"), j.get("loss_section_us"))
PY
