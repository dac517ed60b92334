#!/bin/bash
# r04i: where phase clock 0 is read: kernel entry / feature staging issued / behind the staging barrier
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; mkdir -p gpurun_out/r04i
for lib in abl1 ablate abl2; do
  for thr in 0 448; do
    echo "=== $lib thread $thr"
    BNF_ABLATE=$(( thr * 256 )) BNF_LIB=$ROOT/ab/libbnf_$lib.so BNF_PHASE_PROF=panel_fwd_bwd timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline 2>&1 | grep "phase clocks" | sed 's/.*total/total/'
  done
done 2>&1 | tee gpurun_out/r04i/phase_clocks.txt
