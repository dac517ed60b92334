#!/bin/bash
# Stall / utilisation counters of every kernel of the bench command, a few per pass
# (counter-only passes: --pmc with --kernel-trace, nothing else).
# Usage: [CMD='python scripts/profile_config.py "C5/8 wind-like MAP (bf16)"'] bash scripts/gpu_counters.sh <tag> ; passes are the lines
# of PASSES below (PASSn= overrides / empties a pass); CMD defaults to the bench command.
TAG=${1:-ctr}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $line --kernel-trace --output-format csv -d "$OUT/pass$i" -o pmc -- \
     bash -c "cd $ROOT && ${CMD:-python bench.py --steps 3 --warmup 2 --no-cpu-baseline}" > "$OUT/pass$i.json" 2> "$OUT/pass$i.err"
  echo "pass $i ($line) rc=$?"
  find "$OUT/pass$i" -name "*kernel_trace*" -delete
done <<PASSES
${PASS1:-SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM}
${PASS2:-SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL}
${PASS3:-TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE}
${PASS4:-TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL}
${PASS5:-TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ}
PASSES
cd "$ROOT"
python scripts/counter_summary.py "$OUT" | tee "$OUT/counters.md"
