#!/bin/bash
# round 4: the z-march kernels -- parity tests, then norm5 on the sub-volume with (default) and without (PEA_ZMARCH=0) the march
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_zmarch.py -x -q -m gpu 2>&1 | tail -15 || exit 1
echo "== exp_3d_split, march (default)"; timeout -k 10 300 python profiles/exp_3d_split.py 2>&1 | grep -v amdgpu.ids
echo "== exp_3d_split, PEA_ZMARCH=0"; PEA_ZMARCH=0 timeout -k 10 300 python profiles/exp_3d_split.py 2>&1 | grep -v amdgpu.ids
