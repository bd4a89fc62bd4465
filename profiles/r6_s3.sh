#!/bin/bash
# round 6, GPU session 3: sections with the full-resolution forward enqueued first, the 3D cross gradient accumulated in its kernel,
# the finished 3D map; GPU suite; default bench line; timelines
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r6_s3_tests.txt 2>&1; echo "tests rc $?" | tee -a gpurun_out/r6_s3_tests.txt; tail -3 gpurun_out/r6_s3_tests.txt
timeout -k 10 400 python bench.py > gpurun_out/r6_bench_s3.json 2> gpurun_out/r6_bench_s3.err; echo "bench rc $?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_bench_s3.json").read().strip().splitlines()[-1])
for k in ("value", "ms_per_step", "ms_min", "ms_max", "kernel_ms", "loss_section_us", "ac3ac4_section_us"):
    print(k, d.get(k))
print({k: (v.get("ms_per_step"), v.get("frac"), v.get("pair_frac")) for k, v in d["configs"].items()})
PY
bash profiles/r6_section3d_timeline.sh > gpurun_out/r6_section3d_timeline.txt 2>&1; grep -v "simple_timer" gpurun_out/r6_section3d_timeline.txt | tail -90
bash profiles/r5_section_timeline.sh > gpurun_out/r6_section_timeline.txt 2>&1; grep -v "simple_timer" gpurun_out/r6_section_timeline.txt | head -45
