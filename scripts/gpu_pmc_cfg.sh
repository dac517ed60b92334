#!/bin/bash
# HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the other configurations:
# usage: gpu_pmc_cfg.sh TAG C3 [C5 ...]
set -u; ulimit -c 0
TAG=${1:-pmc_cfg}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in "$@"; do
  OUT=$ROOT/gpurun_out/${TAG}_$cfg; mkdir -p "$OUT"
  cd /tmp && export TMPDIR=/tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$OUT/$c" -o pmc -- \
       python "$ROOT/scripts/bench_configs.py" $cfg > "$OUT/$c.json" 2> "$OUT/$c.err"
    echo "$cfg $c rc=$?"
  done
  cd "$ROOT"
  python scripts/pmc_summary.py "$OUT" | tee "$OUT/traffic.md" | cut -c1-200 | head -12
  find "$OUT" -name "*.csv" -size +5M -delete
done
