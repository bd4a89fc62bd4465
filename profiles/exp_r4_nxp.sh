#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
for cfg in c5 c4 c5 c4; do
  timeout -k 10 300 python bench.py --config $cfg --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/nxp_$cfg.json 2> gpurun_out/nxp_$cfg.err || echo "$cfg failed"
  python3 -c "
import json,sys
j=json.loads(open('gpurun_out/nxp_$cfg.json').read().strip().splitlines()[-1]); print('$cfg', j['ms_per_step'], j['kernel_ms'], j['roofline']['frac'], j['roofline']['fwd_plus_bwd_frac'])"
done
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
