#!/bin/bash
# block shapes of the z-fastest tile walk (profiles/exp_3d_walk.py), one process per setting -> gpurun_out/r3u_walk.txt
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
for yx in "4 2" "8 4" "2 2" "4 4" "16 8" "1 1" "8 2" "-1 0" "2 1"; do
  set -- $yx
  PEA_ZBLK_Y=$1 PEA_ZBLK_X=$2 timeout -k 10 200 python3 profiles/exp_3d_walk.py >> gpurun_out/r3u_walk.txt 2>&1 || exit 1
done
cat gpurun_out/r3u_walk.txt
