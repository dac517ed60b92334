#!/bin/bash
# r04q: what k_vi_sample_pack waits for -- two counter passes over the C3/8 step
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
export BNF_LIB=$ROOT/ab/libbnf_${1:-cur}.so
bash scripts/gpu_counters_cmd.sh r04q_a "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" python scripts/bench_configs.py C3 | cut -c1-260
bash scripts/gpu_counters_cmd.sh r04q_b "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_SALU" python scripts/bench_configs.py C3 | cut -c1-260
