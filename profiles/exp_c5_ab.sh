#!/bin/bash
# bash profiles/exp_c5_ab.sh: configs[4] with the producer / consumer f16 backward (PEA_H16_HW=2, default) and with round 3's LDS-DMA
# backward (=1), alternating, same box, same flags
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in 1 2 3 4 5; do for hw in 2 1; do
  PEA_H16_HW=$hw timeout -k 10 120 python3 $ROOT/bench.py --config c5 --steps 200 --warmup 20 --no-cpu-baseline > /tmp/ab.json 2>/tmp/ab.err || { echo FAILED; continue; }
  python3 -c "
import json; j=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); k=j['kernel_ms']; print('PEA_H16_HW=$hw  step %.4f ms  graph %.4f  fwd %.1f us  bwd %.1f us' % (j['ms_per_step'], j['graph_replay_ms'], k['fwd']*1e3, k['bwd']*1e3))"
done; done
