#!/bin/bash
# round 6, GPU session 1: the pruned library's GPU suite, the super-block walk A/B, the default bench line (with `configs`), c4crop
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r6_s1_tests.txt 2>&1; echo "tests rc $?" | tee -a gpurun_out/r6_s1_tests.txt; tail -3 gpurun_out/r6_s1_tests.txt
timeout -k 10 300 python profiles/r6_sup.py > gpurun_out/r6_sup.txt 2>&1; echo "sup rc $?"; tail -32 gpurun_out/r6_sup.txt
timeout -k 10 400 python bench.py > gpurun_out/r6_bench_s1.json 2> gpurun_out/r6_bench_s1.err; echo "bench rc $?"; tail -c 3000 gpurun_out/r6_bench_s1.json
timeout -k 10 200 python bench.py --config c4crop --steps 100 > gpurun_out/r6_c4crop_s1.json 2> gpurun_out/r6_c4crop_s1.err; echo "c4crop rc $?"; tail -c 1500 gpurun_out/r6_c4crop_s1.json
