#!/bin/bash
# Round 4, final sources: HBM-side bytes (FETCH_SIZE / WRITE_SIZE passes) of every bench configuration -> profiles/traffic.json
# (copied to gpurun_out/ so that it comes back from the GPU box), then SQ / TA / TCC counters of the 3D march kernels and the
# FETCH / TCC passes of c3 under the two tile walks.  bash profiles/r4_final_pmc.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out && rm -f profiles/traffic.json
bash profiles/pmc_step.sh r4f g h > gpurun_out/r4f_pmc_c2.txt 2>&1
python3 profiles/make_traffic.py c2 gpurun_out/pmcs_r4f_g gpurun_out/pmcs_r4f_h >> gpurun_out/r4f_pmc_c2.txt 2>&1
echo "c2 done"
for cfg in c1 c1k8 c3 c4 c4n26 c5 c5f32; do
  bash profiles/pmc_cfg.sh r4f_$cfg $cfg g h > gpurun_out/r4f_pmc_$cfg.txt 2>&1
  python3 profiles/make_traffic.py $cfg gpurun_out/pmcc_r4f_${cfg}_g gpurun_out/pmcc_r4f_${cfg}_h >> gpurun_out/r4f_pmc_$cfg.txt 2>&1
  echo "$cfg done"
done
PEA_BENCH_EXTRA="--batch 32" bash profiles/pmc_cfg.sh r4f_c2b32 c2 g h > gpurun_out/r4f_pmc_c2b32.txt 2>&1
python3 profiles/make_traffic.py c2b32 gpurun_out/pmcc_r4f_c2b32_g gpurun_out/pmcc_r4f_c2b32_h >> gpurun_out/r4f_pmc_c2b32.txt 2>&1
echo "c2b32 done"
cp profiles/traffic.json gpurun_out/traffic.json
cat gpurun_out/traffic.json
cat gpurun_out/r4f_pmc_*.txt | grep -v "^{\|^ \|^}" > gpurun_out/r4_pmc_traffic_passes.txt
# the march kernels: where the cycles go
bash profiles/pmc_cfg.sh r4f_c4x c4 a b c f > gpurun_out/r4_c4_pmc_sq_ta_tcc.txt 2>&1
PEA_ZMARCH=0 bash profiles/pmc_cfg.sh r4f_c4old c4 f g h > gpurun_out/r4_c4_zmarch0_pmc.txt 2>&1
# c3 under the row-major walk and under strips of six tiles
bash profiles/pmc_cfg.sh r4f_c3w0 c3 f g > gpurun_out/r4_c3_walk_pmc.txt 2>&1
PEA_WALK2D=6 bash profiles/pmc_cfg.sh r4f_c3w6 c3 f g >> gpurun_out/r4_c3_walk_pmc.txt 2>&1
echo "pmc extras done"
