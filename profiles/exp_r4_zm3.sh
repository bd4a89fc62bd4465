#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT && mkdir -p gpurun_out
python profiles/exp_zm.py 2>&1 | grep -v amdgpu.ids
export ITERS=3 CASES=fwd,bwd
bash profiles/pmc_script.sh zm_a profiles/exp_zm.py "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU"
bash profiles/pmc_script.sh zm_b profiles/exp_zm.py "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS"
bash profiles/pmc_script.sh zm_g profiles/exp_zm.py "FETCH_SIZE"
bash profiles/pmc_script.sh zm_h profiles/exp_zm.py "WRITE_SIZE"
bash profiles/pmc_script.sh zm_f profiles/exp_zm.py "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"
bash profiles/pmc_script.sh zm_c profiles/exp_zm.py "TA_BUSY_avr TA_BUFFER_TOTAL_CYCLES_sum"
