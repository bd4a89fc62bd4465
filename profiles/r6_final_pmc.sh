#!/bin/bash
# Round 6, final sources: L2-miss traffic (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, as MI355X_MICROARCH.md prescribes) of
# every bench configuration -> profiles/traffic.json.  The passes write gpurun_out/traffic_r6.json; it replaces the tracked record only
# when EVERY pass and every make_traffic.py call succeeded (round-5 advice: the old script deleted the record first and checked nothing).
# bash profiles/r6_final_pmc.sh [configs...]
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
export PEA_TRAFFIC_OUT=$ROOT/gpurun_out/traffic_r6.json
rm -f $PEA_TRAFFIC_OUT
CFGS=${@:-c1 c3 c4 c4crop c5 c2ema c1k8 c3crop c4n26 c5f32 c3ema c4ema c5ema c4r6}
fail=0
bash profiles/pmc_step.sh r6f g h > gpurun_out/r6f_pmc_c2.txt 2>&1 || fail=1
grep -q "FAILED/timeout" gpurun_out/r6f_pmc_c2.txt && fail=1
python3 profiles/make_traffic.py c2 gpurun_out/pmcs_r6f_g gpurun_out/pmcs_r6f_h >> gpurun_out/r6f_pmc_c2.txt 2>&1 || fail=1
echo "c2 done (fail=$fail)"
for cfg in $CFGS; do
  bash profiles/pmc_cfg.sh r6f_$cfg $cfg g h > gpurun_out/r6f_pmc_$cfg.txt 2>&1 || fail=1
  grep -q "FAILED/timeout" gpurun_out/r6f_pmc_$cfg.txt && fail=1
  python3 profiles/make_traffic.py $cfg gpurun_out/pmcc_r6f_${cfg}_g gpurun_out/pmcc_r6f_${cfg}_h >> gpurun_out/r6f_pmc_$cfg.txt 2>&1 || fail=1
  echo "$cfg done (fail=$fail)"
done
cat gpurun_out/r6f_pmc_*.txt | grep -v "^{\|^ \|^}" > gpurun_out/r6_pmc_traffic_passes.txt
if [ $fail = 0 ]; then cp $PEA_TRAFFIC_OUT profiles/traffic.json; echo "traffic.json replaced"; else echo "a pass FAILED: profiles/traffic.json left as it was"; fi
cat $PEA_TRAFFIC_OUT
