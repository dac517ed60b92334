#!/bin/bash
# r03 visit S: 16-byte accesses in the optimiser / sampling / packing kernels (C3 A/B against HEAD~), W = 256 panel
# variant with 128-row panels and two workgroups per CU (BNF_PANEL_RT2, C5 A/B), parity tests of both
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03s}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest.txt" | tail -5 | cut -c1-220
echo "== pytest W=256 panel tests with 128-row panels"; BNF_PANEL_RT2=1 timeout 600 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_rt2.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest_rt2.txt" | tail -5 | cut -c1-220
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C3
  cfg new C3
done 2>&1 | tee "$OUT/ab_c3.txt"
for rep in 1 2; do
  BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C5
  cfg new C5
  BNF_PANEL_RT2=1 cfg new_rt2 C5
done 2>&1 | tee "$OUT/ab_c5.txt"
BNF_LIB=$ROOT/ab/libbnf_head.so cfg head C4; cfg new C4
echo "== C3 per-kernel"; timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
echo "== C5 per-kernel"; BNF_PANEL_RT2=1 timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | tee "$OUT/c5_profile_rt2.txt"
echo "== C2"; VARIANTS="head:ab/libbnf_head.so new:" REPS=2 STEPS=20 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
