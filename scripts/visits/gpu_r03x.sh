#!/bin/bash
# r03 visit X: W = 512 with 64-row panels, 128 registers per wave, two workgroups per CU (BNF_PANEL_Q=1)
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03x}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest (Q)"; BNF_PANEL_NO_H0L=1 BNF_PANEL_Q=1 timeout 400 python -m pytest tests/test_gpu_panel.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > "$OUT/pytest_q.txt" 2>&1; echo "rc=$?"; grep -E "passed|failed|Error" "$OUT/pytest_q.txt" | tail -12 | cut -c1-220
b() { timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))"; }
for rep in 1 2; do
  b base
  BNF_PANEL_NO_H0L=1 b base_noh0l
  BNF_PANEL_NO_H0L=1 BNF_PANEL_Q=1 b q
done 2>&1 | tee "$OUT/ab_c2.txt"
