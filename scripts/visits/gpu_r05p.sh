#!/bin/bash
# r05p: k_featurize with an odd-dword tile pitch, paired 2-byte stores and 16-byte seasonal-table loads:
# full GPU suite with the in-tree build, then C5 / C2 per-kernel tables, base (HEAD) against the new build, both dtypes
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05p}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-200 | head -20 | tee "$OUT/pytest.txt"
for rep in 1 2; do
  for v in base feat; do for dt in bf16 fp8; do
    echo "== $v $dt"
    BNF_BENCH_DTYPE=$dt BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 300 python scripts/bench_configs.py C5 2>/dev/null | tail -1 | cut -c100-200
    BNF_BENCH_DTYPE=$dt BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 200 python scripts/profile_config.py "C5/8 wind-like MAP (bf16)" 2>/dev/null | grep -E "featurize|panel"
  done; done
done 2>&1 | tee "$OUT/c5.txt"
for rep in 1 2; do for v in base feat; do
  BNF_LIB=$ROOT/ab/libbnf_$v.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$v.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['value']))"; grep -E "featurize" "$OUT/bench_$v.err" | head -2
done; done 2>&1 | tee "$OUT/c2.txt"
