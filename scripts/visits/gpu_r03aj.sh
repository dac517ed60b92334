#!/bin/bash
# r03 visit AJ: accuracy of the fp32 engine with libm-class vs hardware exp2 / rcp activation: smoke errors, golden
# reproduction errors (printed by the estimator tests with -s), fp32 parity sweep
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03aj}; mkdir -p "$OUT"; cd "$ROOT"
for v in libm fast; do
  if [ $v = fast ]; then export BNF_LIB=$ROOT/ab/libbnf_fp32fast.so; else unset BNF_LIB; fi
  echo "== $v"
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep "\[smoke\]"
  timeout 300 python scripts/fp32_accuracy.py 2>/dev/null
done 2>&1 | tee "$OUT/accuracy.txt"
