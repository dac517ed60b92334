#!/bin/bash
# r03 visit E: panel tests, same-box A/B (baseline kernels / visit-D build / HEAD), phase clocks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/${1:-r03e}; mkdir -p "$OUT"; cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== pytest"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x > "$OUT/pytest.txt" 2>&1; echo "pytest rc=$?"; tail -5 "$OUT/pytest.txt"
echo "== A/B C2"; VARIANTS="${VARIANTS:-prev:ab/libbnf_prev.so d:ab/libbnf_d.so new:}" STEPS=30 REPS=3 bash scripts/gpu_abn.sh 2>&1 | tee "$OUT/ab.txt"
