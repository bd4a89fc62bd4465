#!/bin/bash
# round 6, GPU session 2: march backward (16-byte g prefetch, quad-transposed stores) and f16 backward (8-byte stores) A/B; 3D section timeline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r6_s2_tests.txt 2>&1; echo "tests rc $?" | tee -a gpurun_out/r6_s2_tests.txt; tail -3 gpurun_out/r6_s2_tests.txt
for r in 1 2; do
  CASES=bwd VARIANTS=old,gq,qs VCASES=bwd AB=0 ITERS=10 timeout -k 10 200 python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r6_zm_bwd_ab.txt
done
for r in 1 2; do
  for q in 0 1; do
    PEA_HQ_QST=$q timeout -k 10 200 python bench.py --config c5 --steps 100 --no-cpu-baseline > gpurun_out/r6_c5_qst${q}_$r.json 2>/dev/null
    python - $q $r <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r6_c5_qst%s_%s.json" % (sys.argv[1], sys.argv[2])).read().strip().splitlines()[-1])
print("c5 PEA_HQ_QST=%s run %s: step %.5f ms (min %.5f max %.5f) kernels %s" % (sys.argv[1], sys.argv[2], d["ms_per_step"], d["ms_min"], d["ms_max"], d["kernel_ms"]))
PY
  done
done 2>&1 | tee gpurun_out/r6_c5_qst_ab.txt
bash profiles/r6_section3d_timeline.sh > gpurun_out/r6_section3d_timeline.txt 2>&1; tail -70 gpurun_out/r6_section3d_timeline.txt
