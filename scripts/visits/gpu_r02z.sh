#!/bin/bash
# r02 final visit: full -m gpu suite, smoke, bench (+cpu_baseline), rocprof kernel stats, PMC traffic,
# MFMA / VALU counters, per-config benches C3 / C4 / C5 (per-GPU share)
set -u
TAG=${1:-r02z}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; cd "$ROOT"
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -vE "^(RCCL|HIP|ROCm|Hostname|Librccl)" > "$OUT/pytest_gpu.txt"; tail -3 "$OUT/pytest_gpu.txt"
bash scripts/gpu_r02n.sh "$TAG"
for c in C3 C4 C5; do timeout 600 python scripts/bench_configs.py $c 2>/dev/null | tail -1; done > "$OUT/configs_bench.jsonl"; cat "$OUT/configs_bench.jsonl" | cut -c1-400
