#!/bin/bash
# r04y: two engine handles of 32 members on ONE GPU, each on its own pair of CU-masked streams (BNF_CU_MEM_EIGHTHS=m:
# m/8 of the CUs for the weight-gradient kernels, the rest for the other kernels) against one handle of 64
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04y; mkdir -p "$OUT"; cd "$ROOT"
export BNF_LIB=$ROOT/ab/libbnf_cumask.so
for rep in 1 2; do
  echo "-- no masks"; timeout 300 python scripts/two_handles_probe.py 1 2 2>&1 | grep "G="
  for m in 1 2 3; do echo "-- BNF_CU_MEM_EIGHTHS=$m"; BNF_CU_MEM_EIGHTHS=$m timeout 300 python scripts/two_handles_probe.py 1 2 4 2>&1 | grep "G=\|rror" | head -5; done
done 2>&1 | tee "$OUT/ab.txt"
