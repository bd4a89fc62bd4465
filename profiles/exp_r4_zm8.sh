#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
VARIANTS=bf8,bf4 python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
for cfg in c1 c1k8; do
  timeout -k 10 300 python bench.py --config $cfg --steps 200 --warmup 20 > gpurun_out/r4x_${cfg}.json 2> gpurun_out/r4x_${cfg}.err || { echo "$cfg failed"; tail -5 gpurun_out/r4x_${cfg}.err; }
  tail -c 1500 gpurun_out/r4x_${cfg}.json; echo
done
timeout -k 10 400 python bench.py --batch 32 --steps 50 --warmup 10 --no-train --no-section > gpurun_out/r4x_b32.json 2> gpurun_out/r4x_b32.err || { echo "b32 failed"; tail -5 gpurun_out/r4x_b32.err; }
tail -c 2500 gpurun_out/r4x_b32.json; echo
