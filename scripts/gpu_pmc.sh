#!/bin/bash
# HBM traffic of every kernel of the bench command, from PMC counters, one counter per pass
# (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit in one pass).
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o pmc -- \
     python "$ROOT/bench.py" --steps 4 --warmup 2 --no-cpu-baseline > "$OUT/$c.json" 2> "$OUT/$c.err"
  echo "$c rc=$?"
done
cd "$ROOT"
find "$OUT" -name "*.csv" | head; 
python scripts/pmc_summary.py "$OUT" | tee "$OUT/traffic.md"
