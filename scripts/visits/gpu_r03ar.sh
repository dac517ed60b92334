#!/bin/bash
# r03 visit AR: weight-ring depth of the panel contraction (fragments in flight per stream): 2 / 4 (built in) / 8
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03ar}; mkdir -p "$OUT"; cd "$ROOT"
for v in pd2 pd8; do echo "== pytest $v"; BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 300 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider 2>&1 | tail -1; done
VARIANTS="pd4: pd2:ab/libbnf_pd2.so pd8:ab/libbnf_pd8.so" REPS=3 STEPS=40 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab_c2.txt"
