#!/bin/bash
# bash profiles/exp_hq_abl.sh: configs[4] (bench.py --config c5) on the product library and on the diagnostic builds of
# pea_k_xdma_hq.hip (profiles/build_variant_tu.sh pea_k_xdma_hq hq_<X> -DPEA_ABL_HQ_<X>) -- what each phase of the register-staged
# f16 kernels costs.  One line per build: kernel_ms fwd / bwd.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
CSRC=$ROOT/pixel-embedded-affinity_amd/csrc
for tag in "" hq_NOSTORE hq_NOLOAD hq_NOGATHER hq_NOWRITE hq_NOSTORE_NOGATHER hq_NOLOAD_NOWRITE ""; do
  lib=$CSRC/libpea_hip${tag:+_$tag}.so
  [ -f $lib ] || continue
  PEA_HIP_LIB=$lib timeout -k 10 120 python3 $ROOT/bench.py --config c5 --steps 60 --warmup 10 --no-cpu-baseline > /tmp/abl.json 2>/tmp/abl.err || { echo "$tag FAILED"; tail -3 /tmp/abl.err; continue; }
  python3 -c "
import json; j=json.loads(open('/tmp/abl.json').read().strip().splitlines()[-1]); print('%-22s' % '${tag:-full}', j['kernel_ms'])"
done
