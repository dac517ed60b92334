#!/bin/bash
# r03 visit N: DEEP template + ring split-K plan: tests, C3 over BNF_RING_SPLITK, C2 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03n}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest all"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "rc=$?"; tail -6 "$OUT/pytest.txt"
cfg() { timeout 300 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  BNF_RING_SPLITK=1 cfg sk1 C3
  BNF_RING_SPLITK=2 cfg sk2 C3
  BNF_RING_SPLITK=4 cfg sk4 C3
  cfg auto C3
done 2>&1 | tee "$OUT/ab_c3.txt"
echo "== C3 per-kernel"; timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
echo "== C2"; VARIANTS="d:ab/libbnf_d.so new:" STEPS=30 REPS=3 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
