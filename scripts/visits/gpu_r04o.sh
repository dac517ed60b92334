#!/bin/bash
# r04o: the whole engine compiled WITHOUT packed f32 VALU instructions (-target-feature -packed-fp32-ops)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04o; mkdir -p "$OUT"; cd "$ROOT"
BNF_LIB=$ROOT/ab/libbnf_nopk.so timeout 900 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2
one() {  # name lib
  local name=$1 lib=$2
  BNF_LIB=$lib timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$name.err" | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', 'ms/step', round(d['ms_per_step'],4), 'panel_us', round(d['roofline']['avg_launch_us'],1))"
  grep "\[bench\]" "$OUT/bench_$name.err" | head -6 | awk '{printf "   %s %s us", $2, $4} END {print ""}'
}
for rep in 1 2 3; do
  one cur $ROOT/bayesnf_amd/libbnf_hip.so
  one nopk $ROOT/ab/libbnf_nopk.so
done 2>&1 | tee "$OUT/ab.txt"
