#!/bin/bash
# r04j: old knobs re-checked on the round-4 kernel: scheduling fences every second group, temporal panel stores,
# static priority for the second-dispatched waves, no contraction priority
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04j; mkdir -p "$OUT"; cd "$ROOT"
one() {  # name lib
  local name=$1 lib=$2
  BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2> /dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
}
for rep in 1 2 3; do
  one cur $ROOT/bayesnf_amd/libbnf_hip.so
  for v in fence2 nt0 prio1 cprio0; do one $v $ROOT/ab/libbnf_$v.so; done
done 2>&1 | tee "$OUT/ab.txt"
