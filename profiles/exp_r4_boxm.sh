#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_zmarch.py -x -q -m gpu -k box_march 2>&1 | tail -15
STENCIL=n26 CASES=fwd,bwd timeout -k 10 300 python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
