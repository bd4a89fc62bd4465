#!/bin/bash
# bash profiles/exp_skew2.sh: PEA_SKEW sweep on the headline (two workgroups per CU in the backward, three in the forward)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
one() {
  PEA_SKEW=$2 PEA_SKEW_SLOTS=$3 PEA_SKEW_MODE=$4 timeout -k 10 120 python3 $ROOT/bench.py --config $1 --steps 100 --warmup 20 --no-cpu-baseline --no-train --no-section > /tmp/sk.json 2>/tmp/sk.err || { echo "$* FAILED"; tail -3 /tmp/sk.err; return; }
  python3 -c "
import json; j=json.loads(open('/tmp/sk.json').read().strip().splitlines()[-1]); k=j['kernel_ms']; print('%-18s' % '$*', j['ms_per_step'], k['fwd'], k['bwd'], k.get('infer'))"
}
for rep in 1 2; do for sk in 0 3 6 9 12 18; do one c2 $sk 2 0; done; done
for sk in 4 8; do one c2 $sk 3 0; done
one c3 0 2 0; one c3 12 2 0; one c3 24 2 0
