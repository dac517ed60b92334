#!/bin/bash
# r03 visit M: depth-generic row-panel kernel -- tests, then C3 A/B (panel vs layer pipeline of the same build, and vs prev), C2 check
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03m}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_panel.txt" 2>&1; echo "rc=$?"; tail -12 "$OUT/pytest_panel.txt"
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; tail -8 "$OUT/pytest.txt"
cfg() { timeout 300 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_prev.so cfg prev C3
  BNF_PIPELINE=0 cfg panel C3
done 2>&1 | tee "$OUT/ab_c3.txt"
echo "== C3 per-kernel"; timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
echo "== C2"; VARIANTS="d:ab/libbnf_d.so new:" STEPS=30 REPS=2 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
