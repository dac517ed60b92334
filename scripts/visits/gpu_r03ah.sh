#!/bin/bash
# r03 visit AH: k_featurize with 32 seasonal-table entries in flight per wait: tests, C5 / C2 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03ah}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|^FAILED|^E  " "$OUT/pytest.txt" | tail -8 | cut -c1-250
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C5; cfg new C5; done 2>&1 | tee "$OUT/ab_c5.txt"
echo "== C5 per-kernel"; timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | tee "$OUT/c5_profile.txt"
echo "== C2"; VARIANTS="head:ab/libbnf_head.so new:" REPS=2 STEPS=20 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
