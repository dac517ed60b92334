#!/bin/bash
# r03 visit J: tests + C3 A/B (noise recovered from the stored samples in k_vi_adam)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03j}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -6 "$OUT/pytest.txt"
for rep in 1 2 3; do
  for v in prev new; do
    if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi
    timeout 300 python scripts/bench_configs.py C3 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$v', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1))"
  done
done 2>&1 | tee "$OUT/ab_c3.txt"
