#!/bin/bash
# r03 visit P: width-1024 panel variant -- tests, C4 A/B (prev / new with and without the LDS feature panel), C2 / C3 check
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03p}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_panel.txt" 2>&1; echo "rc=$?"; tail -25 "$OUT/pytest_panel.txt" | cut -c1-220
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; tail -8 "$OUT/pytest.txt" | cut -c1-220
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_prev.so cfg prev C4
  cfg new C4
  BNF_PANEL_NO_H0L=1 cfg new_noh0l C4
done 2>&1 | tee "$OUT/ab_c4.txt"
echo "== C4 per-kernel"; timeout 300 python scripts/profile_config.py "C4/8 synthetic minibatch MLE" 2>/dev/null | tee "$OUT/c4_profile.txt"
cfg new C3; cfg new C5
echo "== C2"; VARIANTS="d:ab/libbnf_d.so new:" STEPS=30 REPS=2 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
