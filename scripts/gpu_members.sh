#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"
for m in ${MEMS:-1 2 3 4 8 16 64}; do
  echo "=== members=$m"
  timeout 300 python bench.py --steps 10 --warmup 3 --profile-all --no-cpu-baseline --members-per-gpu $m 2>&1 | grep -E "\[bench\]|ms_per_step" | sed 's/"config".*//' 
done
