#!/bin/bash
# r03 visit V: two half panels per workgroup one barrier apart (BNF_PANEL_PP=1): parity tests, C2 / C3 A/B
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03v}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest (PP)"; BNF_PANEL_PP=1 timeout 400 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_pp.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest_pp.txt" | tail -12 | cut -c1-220
b() { timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))"; }
for rep in 1 2 3; do
  b base
  BNF_PANEL_PP=1 b pp
  BNF_PANEL_NO_H0L=1 b base_noh0l
  BNF_PANEL_PP=1 BNF_PANEL_NO_H0L=1 b pp_noh0l
done 2>&1 | tee "$OUT/ab_c2.txt"
cfg() { timeout 400 python scripts/bench_configs.py $2 2>/dev/null | python -c "import sys,json
for l in sys.stdin:
  d=json.loads(l); print('$1', d['config'][:5], round(d['member_steps_per_s'],1), round(d['algorithmic_tflops'],1), round(d['final_loss_mean'],1))"; }
for rep in 1 2; do
  cfg base C3
  BNF_PANEL_PP=1 cfg pp C3
done 2>&1 | tee "$OUT/ab_c3.txt"
