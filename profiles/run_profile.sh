#!/bin/bash
# Usage (on the GPU box, from the repo root):  bash profiles/run_profile.sh <tag> [bench args...]
# Produces gpurun_out/prof_<tag>/{stats,pmc_fetch,pmc_write,pmc_l2}/...csv ; copy the summaries into profiles/.
set -e
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-section --no-configs $@"  # (--no-configs: the other configurations launch kernels of the same names)
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o r -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o r -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o r -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_write.log 2>&1
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_l2 -o r -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_l2.log 2>&1 || true
python3 $ROOT/profiles/summarize.py $OUT > $OUT/summary.txt 2>&1 || true
cat $OUT/summary.txt
