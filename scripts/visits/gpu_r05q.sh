#!/bin/bash
# r05q: which of r05p's three k_featurize changes costs what (C5/8, bf16): variants of ab/
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05q}; shift; mkdir -p "$OUT"; cd "$ROOT"
for rep in 1 2; do for v in "$@"; do
  echo "== $v $(BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | grep -E "featurize")"
done; done 2>&1 | tee "$OUT/feat_ab.txt"
