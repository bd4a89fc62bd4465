#!/bin/bash
# round 4, first GPU call: baselines on this box + two design inputs (1 workgroup per CU on the 3D volume; the 2D strip walk)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
echo "== exp_3d_split, two workgroups per CU (default)"; timeout -k 10 300 python profiles/exp_3d_split.py 2>&1 | grep -v amdgpu.ids
echo "== exp_3d_split, PEA_LDS_PAD=40000 (one workgroup per CU)"; PEA_LDS_PAD=40000 timeout -k 10 300 python profiles/exp_3d_split.py 2>&1 | grep -v amdgpu.ids
bash profiles/exp_walk2d.sh
echo "== cross parity tests under PEA_WALK2D=6"; PEA_WALK2D=6 timeout -k 10 600 python -m pytest tests/test_gpu_cross.py -x -q -m gpu 2>&1 | tail -3
