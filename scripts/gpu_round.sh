#!/bin/bash
# One GPU-box visit: diagnostics, GPU tests, bench, rocprof kernel trace.
# Usage (from the repo root on the GPU box): bash scripts/gpu_round.sh [tag]
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== rocm-smi" ; rocm-smi --showproductname 2>/dev/null | head -8
echo "== diag";   timeout 900 python tests/gpu_diag.py > "$OUT/diag.txt" 2>&1; echo "diag rc=$?"
echo "== pytest"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"
tail -5 "$OUT/pytest.txt"
echo "== smoke";  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1; echo "smoke rc=$?"
echo "== bench";  timeout 900 python bench.py --steps 20 --warmup 3 --profile-all > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "bench rc=$?"
cat "$OUT/bench.json"; tail -20 "$OUT/bench.err"
echo "== rocprof"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o bench -- python "$ROOT/bench.py" --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/rocprof_bench.json" 2> "$OUT/rocprof.err"; echo "rocprof rc=$?"
cd "$ROOT"
find "$OUT/prof" -name "*stats*" | head
for f in $(find "$OUT/prof" -name "*kernel_stats*.csv" | head -1); do head -25 "$f"; done
# keep the merge-back small
find "$OUT/prof" -name "*kernel_trace*" -size +20M -delete
du -sh "$OUT"
