#!/bin/bash
# phase clocks of the panel kernel (ABLATE build in ab/libbnf_ablate.so) seen by several waves, and under ablation masks
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for thr in ${THREADS:-0 64 192 448}; do
  for m in ${ABL_MASKS:-0}; do
    echo "=== thread $thr ablate $m"
    BNF_ABLATE=$(( m + thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_ablate.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/'
  done
done
