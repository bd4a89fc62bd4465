#!/bin/bash
# Round 5 (verdict item 6): SQ / LDS / VMEM / TCC counters of the labels-in training step beside the tensor forward it replaces --
# k_fwd_xdma<.., LAB> + k_bwd_xdma (two launches), k_fused_labels (one launch), and the plain k_fwd_xdma with t / w / m tensors.
#   bash profiles/r5_labels_pmc.sh  -> gpurun_out/r5_labels_pmc.txt
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
for w in labels2 fwd_ex labels; do
  rm -f gpurun_out/pmc_r5lab_$w/summary.txt  # (round-5 advice: a failed pass must not reprint the previous run's summary)
  bash profiles/pmc_kernel.sh r5lab_$w $w > gpurun_out/pmc_r5lab_$w.log 2>&1 || echo "#### pmc_kernel.sh $w FAILED (gpurun_out/pmc_r5lab_$w.log)"
  echo "#### one_kernel.py $w"; cat gpurun_out/pmc_r5lab_$w/summary.txt 2>/dev/null || echo "(no summary: the pass failed)"
done > gpurun_out/r5_labels_pmc.txt 2>&1
echo done
