#!/bin/bash
# r05s: k_featurize by column ranges (n_split): full GPU suite with the in-tree build, C5/8 per-kernel numbers of the variants
# in both dtypes, the kernel at C2 (bf16: one range; fp32: two) with BNF_FEAT_SPLIT=1 against the default
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05s}; shift; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-200 | head -20 | tee "$OUT/pytest.txt"
for rep in 1 2; do for v in "$@"; do for dt in bf16 fp8; do
  echo "== $v $dt $(BNF_BENCH_DTYPE=$dt BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/bench_configs.py C5 2>/dev/null | tail -1 | grep -o '"member_steps_per_s": [0-9.]*') $(BNF_BENCH_DTYPE=$dt BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | grep -E "featurize")"
done; done; done 2>&1 | tee "$OUT/feat_ab.txt"
for rep in 1 2; do for sp in 1 0; do for dt in bf16 fp32; do
  BNF_FEAT_SPLIT=$sp python bench.py --dtype $dt --steps 20 --warmup 3 --no-cpu-baseline --profile-all 2> "$OUT/bench_${dt}_$sp.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$sp $dt', round(d['ms_per_step'],4))"; grep -E "featurize" "$OUT/bench_${dt}_$sp.err" | head -1
done; done; done 2>&1 | tee "$OUT/c2.txt"
