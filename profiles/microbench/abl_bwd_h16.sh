#!/bin/bash
# ablations of the f16 projection-first backward -> gpurun_out/r3_abl_bwd_h16.txt
cd "$(dirname "$0")" && mkdir -p ../../gpurun_out && out=../../gpurun_out/r3_abl_bwd_h16.txt && : > $out
for f in "" -DPEA_ABL_H_NOGATHER -DPEA_ABL_H_NODMA -DPEA_ABL_H_NOSTORE -DPEA_ABL_H_NOILV "-DPEA_ABL_H_NOGATHER -DPEA_ABL_H_NOSTORE" \
         "-DPEA_ABL_H_NOGATHER -DPEA_ABL_H_NOSTORE -DPEA_ABL_H_NOILV" "-DPEA_ABL_H_NOGATHER -DPEA_ABL_H_NOSTORE -DPEA_ABL_H_NOILV -DPEA_ABL_H_NODMA"; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 $f -o /tmp/abl_h16 abl_bwd_h16.hip || exit 1
  echo "== ${f:-full}" >> $out
  timeout -k 10 60 /tmp/abl_h16 >> $out 2>&1
done
cat $out
