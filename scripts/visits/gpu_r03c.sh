#!/bin/bash
# r03 visit C: GPU tests, same-box A/B at C2 and on C3 / C4 / C5
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03c}; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -25 "$OUT/pytest.txt"
echo "== A/B C2"; STEPS=30 bash scripts/gpu_ab.sh 2>&1 | tee "$OUT/ab.txt"
echo "== A/B configs"
for rep in 1 2; do
  for v in prev new; do
    if [ $v = prev ]; then export BNF_LIB=$ROOT/ab/libbnf_prev.so; else unset BNF_LIB; fi
    for c in C3 C4 C5; do
      timeout 300 python scripts/bench_configs.py $c 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$v', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1))"
    done
  done
done 2>&1 | tee "$OUT/ab_configs.txt"
