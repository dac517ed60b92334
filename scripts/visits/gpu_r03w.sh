#!/bin/bash
# r03 visit W: PP geometry with / without the one-barrier lag, with / without priority for the contracting wave
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03w}; mkdir -p "$OUT"; cd "$ROOT"
b() { timeout 180 python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4))"; }
for rep in 1 2; do
  b base
  BNF_PANEL_PP=1 b pp_lag1
  BNF_PANEL_PP=1 BNF_PANEL_PP_LAG=0 b pp_lag0
  BNF_PANEL_PP=1 BNF_PANEL_PP_PRIO=1 b pp_lag1_prio
  BNF_PANEL_PP=1 BNF_PANEL_PP_LAG=0 BNF_PANEL_PP_PRIO=1 b pp_lag0_prio
done 2>&1 | tee "$OUT/ab_c2.txt"
