#!/bin/bash
# r03 visit AG: fp32 engine with k_last_bwd<float> at one wave per SIMD's register budget (no spills): tests, fp32 C2 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03ag}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.txt" | tail -8 | cut -c1-250
b() { timeout 300 python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2>"$OUT/err_$1.txt" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), round(d['value']))"; }
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_head.so b head
  b new
done 2>&1 | tee "$OUT/ab_fp32.txt"
grep "\[bench\]" "$OUT/err_head.txt" | head -12; echo; grep "\[bench\]" "$OUT/err_new.txt" | head -12
