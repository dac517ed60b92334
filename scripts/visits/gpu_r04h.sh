#!/bin/bash
# r04h: Adam's moments read non-temporally (they are touched once per step)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04h; mkdir -p "$OUT"; cd "$ROOT"
one() {  # name lib
  local name=$1 lib=$2
  BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -6 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one cur $ROOT/bayesnf_amd/libbnf_hip.so
  one adamnt $ROOT/ab/libbnf_adamnt.so
done 2>&1 | tee "$OUT/ab.txt"
