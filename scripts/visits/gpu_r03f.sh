#!/bin/bash
# r03 visit F: persistent multi-panel workgroups -- panel tests, then same-box A/B over BNF_PANEL_PPW
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03f}; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.txt"
run() { python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1))"; }
for rep in 1 2 3; do
  BNF_LIB=$ROOT/ab/libbnf_d.so run d
  for ppw in 1 2 5 10 20; do BNF_PANEL_PPW=$ppw run ppw$ppw; done
done 2>&1 | tee "$OUT/ab.txt"
