#!/bin/bash
# time the forward with experimental builds (one ablation each)
for tag in BASE NO_NT NO_MASK NO_G NO_RED NO_TW; do
  if [ $tag = BASE ]; then unset PEA_HIP_LIB; else export PEA_HIP_LIB=$PWD/pixel-embedded-affinity_amd/csrc/exp/libpea_hip_$tag.so; fi
  echo "== $tag"; python profiles/time_one.py fwd
done
