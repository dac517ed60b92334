#!/bin/bash
# r05m: 'fp32' = split-bf16 contractions by default (BNF_DTYPE_F32S), 'fp32_exact' the exact chain: full suite, smoke, bench lines
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05m}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-260 | head -30 | tee "$OUT/pytest_gpu.txt"
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee "$OUT/smoke.txt"
for dt in fp32 fp32_exact bf16; do python bench.py --dtype $dt --steps 10 --warmup 2 --no-cpu-baseline --profile-all 2> "$OUT/bench_$dt.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', round(d['ms_per_step'],3), round(d['value']), round(d['algorithmic_tflops'],1), d['final_loss_mean'])"; grep "\[bench\]" "$OUT/bench_$dt.err" | head -9; done 2>&1 | tee "$OUT/bench_dtypes.txt"
echo "== C1"; timeout 300 python scripts/c1_step_time.py 2>/dev/null | tee "$OUT/c1_step_time.txt"
