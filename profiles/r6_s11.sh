#!/bin/bash
# round 6, GPU session 11: the tile-per-plane 3D kernels' block walk on the reference's training crops (B = 2 x 18 x 160 x 160, norm5)
mkdir -p gpurun_out
DIMS=18,160,160 B=2 BLOCKS=4x2,2x2,2x5,5x5,10x5,10x1,1x5,5x1,3x3,4x4,8x2,2x3,5x3,3x5 SUPS=0 ROUNDS=2 ITERS=20 timeout -k 10 600 python profiles/r6_sup.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_blk_c4crop.txt
cat gpurun_out/r6_blk_c4crop.txt | cut -c1-105
