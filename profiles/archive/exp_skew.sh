#!/bin/bash
# bash profiles/exp_skew.sh: the start-skew experiment (PEA_SKEW, pea_xdma.h xdma_tile) on configs[4] and the headline
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
one() {  # config skew slots mode
  PEA_SKEW=$2 PEA_SKEW_SLOTS=$3 PEA_SKEW_MODE=$4 timeout -k 10 120 python3 $ROOT/bench.py --config $1 --steps 60 --warmup 10 --no-cpu-baseline --no-train --no-section > /tmp/sk.json 2>/tmp/sk.err || { echo "$* FAILED"; tail -3 /tmp/sk.err; return; }
  python3 -c "
import json; j=json.loads(open('/tmp/sk.json').read().strip().splitlines()[-1]); print('%-18s' % '$*', j['ms_per_step'], j['kernel_ms'])"
}
for a in "c5 0 4 0" "c5 10 4 0" "c5 10 4 1" "c5 32 2 0" "c5 32 2 1" "c5 0 4 0" "c2 0 3 0" "c2 7 3 0" "c2 7 3 1" "c2 6 2 0" "c2 6 2 1" "c2 0 3 0"; do one $a; done
