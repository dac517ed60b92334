#!/bin/bash
# r03 visit Q: multi-layer weight-gradient launch + VI optimiser keep ranges: tests, C3 / C4 A/B over BNF_WGRAD_NO_MULTI
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03q}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; tail -8 "$OUT/pytest.txt" | cut -c1-220
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_WGRAD_NO_MULTI=1 cfg single C3
  cfg multi C3
done 2>&1 | tee "$OUT/ab_c3.txt"
BNF_WGRAD_NO_MULTI=1 cfg single C4; cfg multi C4
echo "== C3 per-kernel"; timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
