#!/bin/bash
# r05r: k_featurize copy-out without per-piece divisions, four pieces in flight: feature / panel / fp8 tests with the in-tree
# build, then C5/8 per-kernel numbers of the variants in both dtypes
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05r}; shift; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_panel.py tests/test_gpu_fp8.py tests/test_gpu_estimator.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-200 | head -20 | tee "$OUT/pytest.txt"
for rep in 1 2; do for v in "$@"; do for dt in bf16 fp8; do
  echo "== $v $dt $(BNF_BENCH_DTYPE=$dt BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | grep -E "featurize")"
done; done; done 2>&1 | tee "$OUT/feat_ab.txt"
