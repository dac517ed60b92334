#!/bin/bash
# r03 visit T: VI sampling fused into the packing pass + layer-0 weight-gradient split choice: tests, C3 A/B/C
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03t}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest.txt" | tail -8 | cut -c1-220
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C3
  BNF_VI_NO_FUSED_PACK=1 cfg apart C3
  cfg new C3
done 2>&1 | tee "$OUT/ab_c3.txt"
BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C4; cfg new C4
BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C5; cfg new C5
echo "== C3 per-kernel"; timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
echo "== C2"; VARIANTS="head:ab/libbnf_head.so new:" REPS=2 STEPS=20 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
