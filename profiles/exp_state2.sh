#!/bin/bash
# bash profiles/exp_state2.sh: the headline step in N fresh processes with bench.py's gpu_state: which clocks does the slow state show?
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq 1 ${N:-8}); do
  timeout -k 10 120 python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --no-section > /tmp/st.json 2>/tmp/st.err || { echo FAILED; continue; }
  python3 - <<'PY'
import json
j = json.loads(open('/tmp/st.json').read().strip().splitlines()[-1]); k = j['kernel_ms']; g = j.get('gpu_state') or {}
print('step %.4f  fwd %.1f  bwd %.1f  sclk %s  mclk %s  power %s W  T %s / %s C' % (j['ms_per_step'], k['fwd'] * 1e3, k['bwd'] * 1e3, g.get('sclk_mhz'), g.get('mclk_mhz'),
      g.get('power_w'), g.get('temp2_c'), g.get('temp3_c')))
PY
done
