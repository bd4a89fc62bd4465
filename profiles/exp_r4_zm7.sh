#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_zmarch.py -x -q -m gpu 2>&1 | tail -5
PEA_ZM_NB=3 timeout -k 10 600 python -m pytest tests/test_gpu_zmarch.py -x -q -m gpu 2>&1 | tail -3
python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
