#!/bin/bash
# Round 5, final sources: HBM-side bytes (FETCH_SIZE / WRITE_SIZE in separate passes) of every bench configuration -> profiles/traffic.json
# (copied to gpurun_out/ so that it comes back from the GPU box).  bash profiles/r5_final_pmc.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && rm -f profiles/traffic.json
bash profiles/pmc_step.sh r5f g h > gpurun_out/r5f_pmc_c2.txt 2>&1
python3 profiles/make_traffic.py c2 gpurun_out/pmcs_r5f_g gpurun_out/pmcs_r5f_h >> gpurun_out/r5f_pmc_c2.txt 2>&1
echo "c2 done"
for cfg in c1 c1k8 c3 c3crop c4 c4n26 c5 c5f32 c2ema c3ema c4ema c5ema c4r6; do
  bash profiles/pmc_cfg.sh r5f_$cfg $cfg g h > gpurun_out/r5f_pmc_$cfg.txt 2>&1
  python3 profiles/make_traffic.py $cfg gpurun_out/pmcc_r5f_${cfg}_g gpurun_out/pmcc_r5f_${cfg}_h >> gpurun_out/r5f_pmc_$cfg.txt 2>&1
  echo "$cfg done"
done
PEA_BENCH_EXTRA="--batch 32" bash profiles/pmc_cfg.sh r5f_c2b32 c2 g h > gpurun_out/r5f_pmc_c2b32.txt 2>&1
python3 profiles/make_traffic.py c2b32 gpurun_out/pmcc_r5f_c2b32_g gpurun_out/pmcc_r5f_c2b32_h >> gpurun_out/r5f_pmc_c2b32.txt 2>&1
echo "c2b32 done"
cp profiles/traffic.json gpurun_out/traffic.json
cat gpurun_out/traffic.json
cat gpurun_out/r5f_pmc_*.txt | grep -v "^{\|^ \|^}" > gpurun_out/r5_pmc_traffic_passes.txt
# SQ / LDS / TA / TCC counters of the headline forward and backward: what the D = 16 backward waits for
bash profiles/pmc_step.sh r5x a b c f > gpurun_out/r5_c2_pmc_sq_ta_tcc.txt 2>&1
echo "pmc extras done"
