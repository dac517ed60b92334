#!/bin/bash
# panel kernel phase clocks under ablation masks (ABLATE build)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for m in ${ABL_MASKS:-0 1 2 4 3 7}; do
  echo "=== BNF_ABLATE=$m"
  BNF_ABLATE=$m BNF_LIB=$ROOT/ab/libbnf_ablate.so BNF_PIPELINE=3 BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/'
done
