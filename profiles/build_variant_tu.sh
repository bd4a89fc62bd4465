#!/bin/bash
# bash profiles/build_variant_tu.sh <translation unit, e.g. pea_k_xdma_hq> <tag> [-DNAME=value ...]: a diagnostic build of the library
# whose named translation unit is compiled with the given macros -> pixel-embedded-affinity_amd/csrc/libpea_hip_<tag>.so (the other
# objects are the product's).  PEA_HIP_LIB=<that path> python bench.py ... then times it (pixel-embedded-affinity_amd/_lib.py).
TU=$1; TAG=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pixel-embedded-affinity_amd/csrc
mkdir -p $CSRC/build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o $CSRC/build/variants/${TU}_$TAG.o $CSRC/$TU.hip || exit 1
OBJS=$(ls $CSRC/build/*.o | grep -v /$TU.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CSRC/libpea_hip_$TAG.so $OBJS $CSRC/build/variants/${TU}_$TAG.o && echo "built libpea_hip_$TAG.so ($*)"
