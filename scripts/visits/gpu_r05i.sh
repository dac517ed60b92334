#!/bin/bash
# r05i: first visit of the fp8 operand-storage path: its tests, C2 / C5 / C3 in bf16 and fp8 on one box, per-kernel times
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05i}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest fp8"; timeout 900 python -m pytest tests/test_gpu_fp8.py -x -q -p no:cacheprovider 2>&1 | tail -25 | tee "$OUT/pytest_fp8.txt"
echo "== C2 bf16 / fp8"; for rep in 1 2; do for dt in bf16 fp8; do python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dtype $dt --profile-all 2> "$OUT/bench_$dt.err" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$dt', round(d['ms_per_step'],4), round(d['value']), d['roofline']['kernel'], round(d['roofline']['avg_launch_us'],1), 'loss', d['final_loss_mean'])"; grep "\[bench\]" "$OUT/bench_$dt.err" | head -12; done; done 2>&1 | tee "$OUT/c2_dtypes.txt"
echo "== configs"; for c in C5 C3 C4; do for dt in bf16 fp8; do echo -n "$c $dt "; BNF_BENCH_DTYPE=$dt timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-260; done; done | tee "$OUT/configs_dtypes.txt"
