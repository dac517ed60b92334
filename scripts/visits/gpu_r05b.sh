#!/bin/bash
# r05b: phase clocks of the L1T build and of the round-4 kernels, waves 0 / 4 / 7
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r05b; mkdir -p "$OUT"; cd "$ROOT"
for lib in ablate ablprev; do
  for thr in 0 256 448; do
    echo "=== $lib thread $thr"
    BNF_ABLATE=$(( thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_$lib.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/' | tail -1
  done
done | tee "$OUT/phase_clocks.txt"
