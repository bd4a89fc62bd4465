#!/bin/bash
# round 6, GPU session 4: GPU suite on the frozen sources, then the PMC traffic passes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -x -q > gpurun_out/r6_s4_tests.txt 2>&1; rc=$?; echo "tests rc $rc" | tee -a gpurun_out/r6_s4_tests.txt; tail -3 gpurun_out/r6_s4_tests.txt
[ $rc = 0 ] || exit 1
bash profiles/r6_final_pmc.sh 2>&1 | tail -150
