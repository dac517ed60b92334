#!/bin/bash
# phase clocks of ab/libbnf_<lib>.so for the listed threads: LIBS="ablate ..." THREADS="0 448" bash scripts/gpu_clk.sh tag
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-clk}; mkdir -p "$OUT"; cd "$ROOT"
for lib in ${LIBS:-ablate}; do
  for thr in ${THREADS:-0 448}; do
    for m in ${ABL_MASKS:-0}; do
      echo "=== $lib thread $thr ablate $m"
      BNF_ABLATE=$(( m + thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_$lib.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/' | tail -1
    done
  done
done | tee "$OUT/phase_clocks.txt"
