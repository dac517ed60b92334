#!/bin/bash
# r02 visit B: panel kernel parity + A/B bench (layer pipeline vs panel pipeline)
set -u
TAG=${1:-r02b}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest panel"; timeout 900 python -m pytest tests/test_gpu_panel.py -m gpu -q -p no:cacheprovider -x > "$OUT/pytest_panel.txt" 2>&1; echo "pytest rc=$?"
tail -30 "$OUT/pytest_panel.txt"
echo "== bench default"; timeout 600 python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"; echo "rc=$?"
cat "$OUT/bench_default.json"; grep -i "us\b\|ms" "$OUT/bench_default.err" | tail -20
echo "== bench panel"; BNF_PIPELINE=3 timeout 600 python bench.py --steps 20 --warmup 3 --profile-all --no-cpu-baseline > "$OUT/bench_panel.json" 2> "$OUT/bench_panel.err"; echo "rc=$?"
cat "$OUT/bench_panel.json"; grep -i "us\b\|ms" "$OUT/bench_panel.err" | tail -20
