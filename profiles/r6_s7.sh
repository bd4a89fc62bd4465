#!/bin/bash
# round 6, GPU session 7: the GPU suite twice more on a fresh box (flakiness check), smoke(), the driver's exact bench command
set -o pipefail
mkdir -p gpurun_out
for i in 1 2; do timeout -k 10 600 python -m pytest tests -m gpu -q 2>&1 | tail -2; done
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r6_driver_cmd.json 2>/dev/null; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r6_driver_cmd.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["ms_min"], d["ms_max"], d["roofline"]["frac"], d["roofline"]["traffic"], d["loss_section_us"], d["ac3ac4_section_us"])
print({k: (v.get("ms_per_step"), v.get("traffic")) for k, v in d["configs"].items()})
PY
