#!/bin/bash
# r05u: one-launch weight packing of the layers pipeline: full GPU suite, C1-sized step times, C2 fp32 line
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r05u}; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest"; timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-220 | head -30 | tee "$OUT/pytest.txt"
echo "== C1 step time"; for rep in 1 2; do timeout 300 python scripts/c1_step_time.py 2>/dev/null; done | tee "$OUT/c1_step_time.txt"
for rep in 1 2; do python bench.py --dtype fp32 --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32', round(d['ms_per_step'],3), round(d['value']))"; done | tee "$OUT/bench_fp32.txt"
