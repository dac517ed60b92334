#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for m in ${ABL_MASKS:-0 1 2 3 8 11 32 43}; do
  echo "=== BNF_ABLATE=$m"
  BNF_PIPELINE=2 BNF_ABLATE=$m timeout 300 python bench.py --steps 4 --warmup 2 --profile-all --no-cpu-baseline 2>&1 >/dev/null | grep "fused_fwd_bwd"
done
