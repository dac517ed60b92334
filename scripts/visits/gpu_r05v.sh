#!/bin/bash
# r05v: gemm_tn_ring8 on the K = 64 block-scaled fp8 MFMA at unit scales (2 x the rate of the K = 16 form):
# fp8 tests with the in-tree build, then C2 / C3 / C4 / C5 in fp8, variants of ab/
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05v}; shift; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 900 python -m pytest tests/test_gpu_fp8.py -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-220 | head -20 | tee "$OUT/pytest.txt"
for rep in 1 2; do for v in "$@"; do
  BNF_LIB=$ROOT/ab/libbnf_$v.so python bench.py --dtype fp8 --steps 30 --warmup 5 --no-cpu-baseline --profile-all 2> "$OUT/bench_$v.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],4), round(d['value']))"; grep -E "gemm_wgrad " "$OUT/bench_$v.err" | head -1
done; done 2>&1 | tee "$OUT/c2.txt"
for v in "$@"; do for c in C3 C4 C5; do
  echo "$v $c $(BNF_BENCH_DTYPE=fp8 BNF_LIB=$ROOT/ab/libbnf_$v.so timeout 300 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | grep -o '"member_steps_per_s": [0-9.]*')"
done; done 2>&1 | tee "$OUT/configs.txt"
