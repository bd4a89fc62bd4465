#!/bin/bash
# Round 3, final sources: HBM-side bytes (FETCH_SIZE / WRITE_SIZE passes) of every bench configuration -> profiles/traffic.json
# (copied to gpurun_out/ so that it comes back from the GPU box).  bash profiles/r3_final_pmc.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && rm -f profiles/traffic.json
bash profiles/pmc_step.sh r3f g h > gpurun_out/r3f_pmc_c2.txt 2>&1
python3 profiles/make_traffic.py c2 gpurun_out/pmcs_r3f_g gpurun_out/pmcs_r3f_h >> gpurun_out/r3f_pmc_c2.txt 2>&1
for cfg in c3 c4 c4n26 c5 c5f32; do
  bash profiles/pmc_cfg.sh r3f_$cfg $cfg g h > gpurun_out/r3f_pmc_$cfg.txt 2>&1
  python3 profiles/make_traffic.py $cfg gpurun_out/pmcc_r3f_${cfg}_g gpurun_out/pmcc_r3f_${cfg}_h >> gpurun_out/r3f_pmc_$cfg.txt 2>&1
  echo "$cfg done"
done
cp profiles/traffic.json gpurun_out/traffic.json
cat gpurun_out/traffic.json
