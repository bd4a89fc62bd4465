#!/bin/bash
# round 6, GPU session 8b: the march's walk, smaller blocks (8 / 16 tile columns per block: an XCD then works on several super-blocks at once)
mkdir -p gpurun_out
BLOCKS=16x1,8x1,4x2,2x4,2x2,4x1,8x2,4x4 SUPS=1,2,4,8 ROUNDS=1 ITERS=6 timeout -k 10 900 python profiles/r6_sup.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_sup_wide2.txt
sort -k13 -n gpurun_out/r6_sup_wide2.txt | awk '{print $4,$5,$6,$8,$10,$13,$16}' | head -14
