#!/bin/bash
# round 6, GPU session 5: the new march-sized finished-pred test, and a soak of the three randomised sweeps with fresh seeds (600 / 300 / 200 cases)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_zmarch.py -m gpu -x -q -k "finished_pred or section_backward" 2>&1 | tail -3
for s in 601 602; do timeout -k 10 500 python tests/fuzz/fuzz_tiled_vs_direct.py 300 $s 2>&1 | grep -v amdgpu.ids | tail -2; done | tee gpurun_out/r6_fuzz_soak.txt
for s in 611 612; do timeout -k 10 400 python tests/fuzz/fuzz_paths.py 150 $s 2>&1 | grep -v amdgpu.ids | tail -2; done | tee -a gpurun_out/r6_fuzz_soak.txt
timeout -k 10 300 python tests/fuzz/fuzz_formats.py 200 621 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a gpurun_out/r6_fuzz_soak.txt
