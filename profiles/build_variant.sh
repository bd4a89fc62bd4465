#!/bin/bash
# bash profiles/build_variant.sh <tag> [-DNAME=value ...]: a diagnostic build of the library whose z-march translation unit is
# compiled with the given macros -> pixel-embedded-affinity_amd/csrc/libpea_hip_<tag>.so (the other objects are the product's).
# profiles/exp_zm.py VARIANTS=<tag>,<tag> then times the variants beside the product library in ONE process on ONE box.
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/pixel-embedded-affinity_amd/csrc
mkdir -p $CSRC/build/variants
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC "$@" -c -o $CSRC/build/variants/zm_$TAG.o $CSRC/pea_k_zmarch.hip || exit 1
OBJS=$(ls $CSRC/build/*.o | grep -v pea_k_zmarch.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $CSRC/libpea_hip_$TAG.so $OBJS $CSRC/build/variants/zm_$TAG.o && echo "built libpea_hip_$TAG.so ($*)"
