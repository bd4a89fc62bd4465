#!/bin/bash
# bash profiles/exp_hf_abl.sh: configs[4] forward (k_fwd_xdma_h) on the product library and on diagnostic builds with a phase compiled out
# (profiles/build_variant_tu.sh pea_k_xdma_h hf_<X> -DPEA_ABL_HF_<X>; the flags existed in that experiment's pea_xdma_h16.h only)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CSRC=$ROOT/pixel-embedded-affinity_amd/csrc
for tag in "" hf_NODMA hf_NOGATHER hf_NODMA_NOGATHER hf_NOTW hf_NOSTORE hf_NOTW_NOSTORE ""; do
  lib=$CSRC/libpea_hip${tag:+_$tag}.so
  [ -f $lib ] || continue
  PEA_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/bench.py --config c5 --steps 100 --warmup 10 --no-cpu-baseline > /tmp/abl.json 2>/tmp/abl.err || { echo "$tag FAILED"; tail -3 /tmp/abl.err; continue; }
  python3 - "${tag:-full}" <<'PY'
import json, sys
j = json.loads(open('/tmp/abl.json').read().strip().splitlines()[-1]); print('%-22s' % sys.argv[1], j['kernel_ms'])
PY
done
