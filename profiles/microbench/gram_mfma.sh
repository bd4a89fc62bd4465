#!/bin/bash
# the MFMA Gram experiment and its ablations (timing only) -> gpurun_out/r3_gram_mfma.txt
cd "$(dirname "$0")" && mkdir -p ../../gpurun_out && out=../../gpurun_out/r3_gram_mfma.txt && : > $out
for f in "" -DDMA_STAGE "-DDMA_STAGE -DABL_NOSTORE" -DABL_NOSTAGE -DABL_NOMFMA -DABL_NOSTORE "-DABL_NOSTAGE -DABL_NOEPI" "-DABL_NOSTAGE -DABL_NOEPI -DABL_NOWRITE" "-DABL_NOSTAGE -DABL_NOEPI -DABL_NOWRITE -DABL_NOMFMA"; do
  hipcc -O3 --offload-arch=gfx950 $f -o /tmp/gram_mfma gram_mfma.hip || exit 1
  echo "== ${f:-full}" >> $out
  timeout -k 10 120 /tmp/gram_mfma 50 >> $out 2>&1
done
cat $out
