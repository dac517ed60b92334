#!/bin/bash
# r05t: Dense kernels split when packed + a five-instruction operand split in the split-bf16 f32 contractions:
# full GPU suite with the in-tree build, then C2 in fp32 / fp32_exact / bf16 with per-kernel times
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05t}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-220 | head -30 | tee "$OUT/pytest.txt"
for rep in 1 2; do for dt in fp32 fp32_exact bf16; do
  python bench.py --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2> "$OUT/bench_$dt.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', round(d['ms_per_step'],3), round(d['value']))"
  [ $rep = 1 ] && grep -E "^\[bench\] [a-z_0-9]+ +avg" "$OUT/bench_$dt.err" | head -10
done; done 2>&1 | tee "$OUT/bench_dtypes.txt"
