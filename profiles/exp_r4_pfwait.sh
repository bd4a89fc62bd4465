#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
for cfg in c5 c3 c5f32 c5 c3; do
  timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/pfw_$cfg.json 2> gpurun_out/pfw_$cfg.err || echo "$cfg failed"
  python3 -c "
import json,sys
j=json.loads(open('gpurun_out/pfw_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', j['ms_per_step'], j['kernel_ms'])"
done
timeout -k 10 600 python -m pytest tests/test_gpu_cross.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py -x -q -m gpu -k "f16 or config or pf or d32 or d64 or cross" 2>&1 | tail -4
python profiles/exp_ema3d.py 2>&1 | grep -v amdgpu
