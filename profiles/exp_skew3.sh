#!/bin/bash
# bash profiles/exp_skew3.sh: the headline step with and without the start skew of the D = 16 backward, alternating, same flags
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4; do for sk in -1 0; do
  PEA_SKEW=$sk timeout -k 10 120 python3 $ROOT/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-train --no-section $EXTRA > /tmp/sk.json 2>/tmp/sk.err || { echo "FAILED"; tail -3 /tmp/sk.err; continue; }
  python3 -c "
import json; j=json.loads(open('/tmp/sk.json').read().strip().splitlines()[-1]); k=j['kernel_ms']; print('skew %2s' % '$sk', 'step', j['ms_per_step'], 'seed', j['ms_per_step_autograd_seed'], 'graph', j['graph_replay_ms'], 'fwd', k['fwd'], 'bwd', k['bwd'])"
done; done
