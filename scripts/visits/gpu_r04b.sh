#!/bin/bash
# r04b: where do the layer-0 phases of the panel kernel lose their time?  phase clocks under combined ablation masks
# (1 / 16: no layer-0 MFMAs, 2: no activation math, 4: no LDS panel stores, 8: no panel copies to HBM, 32: no feature staging,
#  64: no layer-0 weight / bias loads)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04b; mkdir -p "$OUT"; cd "$ROOT"
THREADS="0 448" ABL_MASKS="${MASKS:-0 2 28 30 32 64 96 126}" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
