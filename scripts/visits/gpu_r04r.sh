#!/bin/bash
# r04r: full GPU suite with the new VI data flow + C3 bench/profile + C2 bench sanity
set -u; ulimit -c 0
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r04r; mkdir -p "$OUT"; cd "$ROOT"
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee "$OUT/pytest_gpu.txt"
for rep in 1 2; do
  timeout 300 python scripts/bench_configs.py C3 2>/dev/null | tail -1 | cut -c1-220
done | tee "$OUT/c3.txt"
timeout 200 python scripts/profile_config.py "C3/8 air_quality-like VI" 2>/dev/null | tee "$OUT/c3_profile.txt"
timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee "$OUT/c2.txt"
