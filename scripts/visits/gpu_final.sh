#!/bin/bash
# last visit of a round: the suite twice (flake screen), smoke, the driver's bench command with the stamped counters in place
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-final}; mkdir -p "$OUT"; cd "$ROOT"
for i in 1 2; do timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED|^E   +(Assertion|assert)" | cut -c1-200 | head -10; done | tee "$OUT/pytest_twice.txt"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee "$OUT/smoke.txt"
python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; cat "$OUT/bench_default.json" | cut -c1-1500
python bench.py --steps 20 --warmup 3 --no-cpu-baseline --dtype fp8 2>/dev/null | cut -c1-300 | tee "$OUT/bench_fp8.json"
