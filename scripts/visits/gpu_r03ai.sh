#!/bin/bash
# r03 visit AI: fp32 engine with the hardware exp2 / rcp activation (build -DBNF_FP32_FAST=1): the whole GPU suite
# against it (fp32 parity bars), fp32 C2 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03ai}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest (fp32 fast activation)"; BNF_LIB=$ROOT/ab/libbnf_fp32fast.so timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest_fast.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED" "$OUT/pytest_fast.txt" | tail -30 | cut -c1-250
b() { timeout 300 python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2>"$OUT/err_$1.txt" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['value']), d['final_loss_mean'])"; }
for rep in 1 2; do
  b libm
  BNF_LIB=$ROOT/ab/libbnf_fp32fast.so b fast
done 2>&1 | tee "$OUT/ab_fp32.txt"
grep "\[bench\]" "$OUT/err_fast.txt" | head -8
