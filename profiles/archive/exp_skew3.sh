#!/bin/bash
# bash profiles/exp_skew3.sh: the headline step with (PEA_SKEW=6, two slots) and without the start skew of the D = 16 backward,
# alternating, same flags; prints bench.py's gpu_state (the card's hwmon clocks / power / temperatures under load) beside each run
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4; do for sk in 6 0; do
  PEA_SKEW=$sk PEA_SKEW_SLOTS=2 timeout -k 10 120 python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --no-section $EXTRA > /tmp/sk.json 2>/tmp/sk.err || { echo "FAILED"; tail -3 /tmp/sk.err; continue; }
  python3 - $sk <<'PY'
import json, sys
j = json.loads(open('/tmp/sk.json').read().strip().splitlines()[-1]); k = j['kernel_ms']
print('skew %s' % sys.argv[1], 'step', j['ms_per_step'], 'fwd', k['fwd'], 'bwd', k['bwd'], j.get('gpu_state'))
PY
done; done
