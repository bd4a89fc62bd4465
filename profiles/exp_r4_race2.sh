#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
N=10 SW="|PEA_ZM_NB=3|PEA_ZBLK_Y=8,PEA_ZBLK_X=4|PEA_ZBLK_Y=4,PEA_ZBLK_X=4" python profiles/exp_zm_race.py 2>&1 | grep -v amdgpu.ids
STENCIL=n26 N=10 SW="|PEA_ZBLK_Y=16,PEA_ZBLK_X=2|PEA_BOXM=0" python profiles/exp_zm_race.py 2>&1 | grep -v amdgpu.ids
CASES=fwd,bwd AB=1 BLOCKS=16x2 python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
STENCIL=n26 CASES=fwd,bwd python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
for cfg in c3 c5 c5f32; do
  timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/race2_$cfg.json 2> gpurun_out/race2_$cfg.err || echo "$cfg failed"
  python3 -c "
import json,sys
j=json.loads(open('gpurun_out/race2_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', j['ms_per_step'], j['kernel_ms'])"
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
