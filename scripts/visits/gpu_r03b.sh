#!/bin/bash
# r03 visit B: GPU tests (incl. fan-out), same-box A/B at C2 and on C3 / C4 / C5, pytest-context flake hunt, phase clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03b}; mkdir -p "$OUT"; cd "$ROOT"
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
unset BNF_LIB
echo "== phase clocks"; THREADS="0 448" bash scripts/gpu_phase_clocks.sh 2>&1 | tee "$OUT/phase_clocks.txt"
echo "== pytest-context flake hunt (tests/test_gpu_parity.py, fresh processes)"
N=${HUNT_N:-12}
for w in 1 2 3 4; do
  ( for i in $(seq $N); do timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done > "$OUT/ctx_$w.txt" 2>&1 ) &
done
wait
cat "$OUT"/ctx_*.txt | sort | uniq -c
